"""Data-parallel host logic on CPU: world_size-2 gloo processes (127.0.0.1)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Log:
    def cprint(self, *a):
        pass


def _worker(rank, world, port, out):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "gfs-3dseg_gws_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gfs3d import dist as gd
    from runs.eval import evaluate_metric_GFS
    res = {}
    # 1. shards are disjoint and cover everything
    res["shard"] = gd.shard_range(37)
    # 2. flat-bucket gradient all-reduce = mean over ranks; afterwards the grads are views of the bucket
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    bucket = gd.FlatGradBucket(lin.parameters())
    x = torch.full((4, 5), float(rank + 1))
    bucket.zero()
    lin(x).sum().backward()
    local = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).clone()
    bucket.all_reduce_mean()
    res["local_grad"], res["mean_grad"] = local.numpy(), bucket.flat.numpy().copy()
    res["views"] = all(p.grad.data_ptr() >= bucket.flat.data_ptr() for p in lin.parameters())
    # 2b. the same with the LAST layer's parameters (first in backward order) exchanged early, under the rest of backward
    early = gd.FlatGradBucket(lin.parameters(), early=lin[1].parameters())
    for _ in range(2):                                   # twice: the hook state must reset with zero()
        early.zero()
        lin(x).sum().backward()
        early.all_reduce_mean()
    res["early_grad"] = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).numpy().copy()
    res["early_launches"] = early.early_launches
    # 3. metric: every rank evaluates its shard of the batches, counters are all-reduced
    g = np.load(os.path.join(here, "golden", "metric.npz"))
    lo, hi = gd.shard_range(len(g["prs"]))
    r = evaluate_metric_GFS(_Log(), list(g["prs"][lo:hi]), list(g["gts"][lo:hi]), list(range(13)), list(g["novel"]),
                            list(g["order"]))
    res["metric"] = np.array(r[:4])
    res["max"] = gd.max_over_ranks(float(rank), "cpu")
    # 4. generate_fake_proto on a block-sharded support half == the single-process result on the whole half
    import random, types
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    gc = np.load(os.path.join(here, "golden", "capl.npz"))
    a = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                              base_widths=[128, 64], output_dim=64, eval_weight=1.2)
    torch.manual_seed(0)
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=a, base_num=7, gp=torch.from_numpy(gc["gp"]), energy=0.9)
    pf = torch.from_numpy(gc["gf_point"]).clone().requires_grad_(True)      # (4,128,256)
    yy = torch.from_numpy(gc["gfp_y"])
    lo, hi = gd.shard_range(4)
    random.seed(0)
    proto, chosen = m.generate_fake_proto(pf[lo:hi], yy[lo:hi], m.main_proto.detach().clone())
    proto.sum().backward()
    res["fake_proto"], res["fake_chosen"] = proto.detach().numpy(), sorted(int(c) for c in chosen)
    res["fake_grad"] = pf.grad.numpy().copy()
    # the differentiable sum is gfs3d.dist's own: a rank that imported torch.distributed.nn aborts at interpreter exit every
    # other run with torch 2.10
    import sys
    res["dist_nn_imported"] = any(k.startswith("torch.distributed.nn") for k in sys.modules)
    # 5. synchronised BatchNorm wiring: BNSync sums fp64 statistics over the ranks; the converted model keeps
    #    its state_dict keys and every EdgeConv block still resolves to the fused HIP path
    from gfs3d import ops
    sync = ops.BNSync.from_group()
    t = torch.full((128,), float(rank + 1), dtype=torch.float64)
    sync.all_reduce(t)
    res["sync_sum"], res["sync_world"] = float(t[0]), sync.world_size
    keys = list(m.state_dict().keys())
    m2 = gd.enable_sync_batchnorm(m)
    res["sync_keys_same"] = list(m2.state_dict().keys()) == keys
    res["sync_attached"] = m2.encoder.bn_sync is not None and all(
        len(blk.fused_parts()) == 4 for blk in m2.encoder.edge_convs)
    # 6. device_kmeans on genuinely SHARDED points: rank 0's k-means++ start is broadcast, the stop
    #    threshold comes from the all-reduced variance, so both ranks run the same iterations and end with the same
    #    centroids - equal to a single-process run over all the points from the same start.  (The assignment step
    #    is the HIP kernel in the product; this CPU logic test injects a torch stand-in for it.)
    from get_basis import device_kmeans
    pts_all, init = _kmeans_data()
    lo, hi = (0, 700) if rank == 0 else (700, 2000)          # unequal shards
    cent, lab, inertia, n_iter = device_kmeans(pts_all[lo:hi], 6, init=init, max_iter=50, tol=1e-4, assign_fn=_torch_assign)
    cent2, _, inertia2, n_iter2 = device_kmeans(pts_all[lo:hi], 6, init=None, max_iter=50, tol=1e-4, seed=3, n_init=2,
                                                assign_fn=_torch_assign)
    res["km"] = (cent.numpy(), lab.numpy(), inertia, n_iter)
    res["km_pp"] = (cent2.numpy(), inertia2, n_iter2)
    #    ... and from a start that leaves a cluster EMPTY: the relocation (sklearn's rule: the globally farthest point) gathers
    #    its candidates from both ranks, so both apply the same move
    far = init.clone()
    far[2] += 100.0
    cent3, lab3, inertia3, n_iter3 = device_kmeans(pts_all[lo:hi], 6, init=far, max_iter=50, tol=1e-4, assign_fn=_torch_assign)
    res["km_far"] = (cent3.numpy(), lab3.numpy(), inertia3, n_iter3)
    # 7. overlap=False: the early slice is exchanged from all_reduce_mean() (the order synchronised BatchNorm needs)
    late = gd.FlatGradBucket(lin.parameters(), early=lin[1].parameters(), overlap=False)
    late.zero()
    lin(x).sum().backward()
    late.all_reduce_mean()
    res["late_grad"] = torch.cat([p.grad.reshape(-1) for p in lin.parameters()]).numpy().copy()
    res["late_launches"] = late.early_launches
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def _kmeans_data():
    g = torch.Generator().manual_seed(4)
    centers = torch.randn(6, 16, generator=g) * 3
    pts = centers[torch.randint(0, 6, (2000,), generator=g)] + 0.4 * torch.randn(2000, 16, generator=g)
    return pts.contiguous(), pts[torch.randperm(2000, generator=g)[:6]].clone()


def _torch_assign(pts, cent):
    """Stand-in for gfs3d.ops.kmeans_assign in CPU logic tests: (label, sums, counts, inertia)."""
    d = torch.cdist(pts.double(), cent.double()) ** 2
    lab = d.argmin(1)
    sums = torch.zeros_like(cent).index_add_(0, lab, pts)
    counts = torch.bincount(lab, minlength=cent.shape[0]).to(pts.dtype)
    return lab.int(), sums, counts, d.gather(1, lab[:, None]).sum().reshape(1)


def test_dp_helpers_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["shard"] == (0, 19) and b["shard"] == (19, 37)
    assert not a["dist_nn_imported"] and not b["dist_nn_imported"]
    mean = (a["local_grad"] + b["local_grad"]) / 2
    assert np.allclose(a["mean_grad"], mean, atol=1e-6) and np.allclose(b["mean_grad"], mean, atol=1e-6)
    assert np.allclose(a["early_grad"], mean, atol=1e-6) and np.allclose(b["early_grad"], mean, atol=1e-6)
    assert a["early_launches"] == 2 and b["early_launches"] == 2        # the overlapped slice ran in both steps
    assert a["views"] and b["views"]
    ref = load_golden("metric")["res"]
    assert np.allclose(a["metric"], ref, atol=1e-12) and np.allclose(b["metric"], ref, atol=1e-12)
    assert a["max"] == 1.0 and b["max"] == 1.0
    assert a["sync_sum"] == b["sync_sum"] == 3.0 and a["sync_world"] == 2
    assert a["sync_keys_same"] and a["sync_attached"] and b["sync_attached"]
    # (6) sharded k-means: identical on both ranks, equal to the single-process run on all points
    from get_basis import device_kmeans
    pts_all, init = _kmeans_data()
    cent, lab, inertia, n_iter = device_kmeans(pts_all, 6, init=init, max_iter=50, tol=1e-4, assign_fn=_torch_assign)
    assert a["km"][3] == b["km"][3] == n_iter
    assert np.array_equal(a["km"][0], b["km"][0]) and np.abs(a["km"][0] - cent.numpy()).max() < 1e-5
    assert np.array_equal(np.concatenate([a["km"][1], b["km"][1]]), lab.numpy())
    assert abs(a["km"][2] - inertia) < 1e-6 * inertia and a["km"][2] == b["km"][2]
    assert np.array_equal(a["km_pp"][0], b["km_pp"][0]) and a["km_pp"][1] == b["km_pp"][1] and a["km_pp"][2] == b["km_pp"][2]
    far = init.clone()
    far[2] += 100.0
    cent, lab, inertia, n_iter = device_kmeans(pts_all, 6, init=far, max_iter=50, tol=1e-4, assign_fn=_torch_assign)
    assert a["km_far"][3] == b["km_far"][3] == n_iter
    assert np.array_equal(a["km_far"][0], b["km_far"][0]) and np.abs(a["km_far"][0] - cent.numpy()).max() < 1e-5
    assert np.array_equal(np.concatenate([a["km_far"][1], b["km_far"][1]]), lab.numpy())
    assert abs(a["km_far"][2] - inertia) < 1e-6 * inertia
    assert len(np.unique(lab.numpy())) == 6                     # the far centre was relocated, not left empty
    assert np.allclose(a["late_grad"], mean, atol=1e-6) and np.allclose(b["late_grad"], mean, atol=1e-6)
    assert a["late_launches"] == 0
    # single-process reference for (4)
    import random, types, sys
    from conftest import PKG
    from model.capl import mpti_net_Point_GeoAsWeight_v2
    gc = load_golden("capl")
    ar = types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=9, dgcnn_k=20,
                               base_widths=[128, 64], output_dim=64, eval_weight=1.2)
    torch.manual_seed(0)
    m = mpti_net_Point_GeoAsWeight_v2(classes=13, args=ar, base_num=7, gp=torch.from_numpy(gc["gp"]), energy=0.9)
    pf = torch.from_numpy(gc["gf_point"]).clone().requires_grad_(True)
    random.seed(0)
    proto, chosen = m.generate_fake_proto(pf, torch.from_numpy(gc["gfp_y"]), m.main_proto.detach().clone())
    proto.sum().backward()
    assert a["fake_chosen"] == b["fake_chosen"] == sorted(int(c) for c in chosen)
    assert np.abs(a["fake_proto"] - proto.detach().numpy()).max() < 1e-6
    assert np.abs(b["fake_proto"] - proto.detach().numpy()).max() < 1e-6
    # every rank back-propagates the same global prototype: summed shard gradients = 2x the single-process one
    full = a["fake_grad"] + b["fake_grad"]
    assert np.abs(full - 2 * pf.grad.numpy()).max() < 1e-5 * max(1.0, np.abs(pf.grad.numpy()).max())


def _run_bench(argv, **env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(GFS3D_BENCH_BACKEND="gloo", GFS3D_BENCH_TIMEOUT="240", **env)
    for attempt in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, env=e, capture_output=True, text=True, timeout=300)
        # the launcher probes a free rendezvous port and rank 0 binds it a moment later: another process of the box can take it in
        # between (seen once in ~40 runs) - one retry on a new port, only for that failure
        if r.returncode == 0 or "GFS3D_BENCH_DRY_FAIL_RANK" in env or not any(k in r.stderr for k in ("EADDRINUSE", "address already in use", "Address already in use")):
            break
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, [json.loads(ln) for ln in lines], r.stderr


def test_bench_gpus_8_launches_eight_ranks_by_itself():
    """a plain `python bench.py --gpus 8` (no torchrun) is an 8-rank job: the parent spawns the ranks,
    rank 0's single JSON line comes back, n_gpus = ranks in the process group = 8.  --dry: bucket + collectives, no kernels."""
    rc, lines, err = _run_bench(["--gpus", "8", "--dry", "--steps", "2"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    ln = lines[0]
    assert ln["n_gpus"] == 8 and ln["rccl_ranks"] == 8 and len(ln["devices"]) == 8 and ln["collectives_ok"]
    assert ln["bucket_elems"] == 261504                    # the DGCNN backbone's parameters


@pytest.mark.parametrize("argv,bucket,overlap,sync_layers", [
    (["--gpus", "8", "--workload", "c4"], 404544, True, 0),
    (["--gpus", "2", "--workload", "c4", "--sync-bn"], 404544, False, 11),
    (["--gpus", "2", "--sync-bn"], 261504, False, 8)])
def test_bench_dry_rehearses_what_a_scaling_run_launches(argv, bucket, overlap, sync_layers):
    """the launcher with `--workload c4` (the CAPL model's bucket: 404 544 gradients, everything outside the
    EdgeConv blocks in the early slice; capl.forward's own data-path collectives - label-presence MAX and the fake-novel
    (sum, count) all-reduce of generate_fake_proto - every rank ending with the same classes and prototypes) and with
    `--sync-bn` (the BatchNorm phase all-reduces on the bucket's communicator, hence NO overlapped early slice).  Rank 0's line
    carries every rank's own time per step (min / max): a straggler is visible."""
    rc, lines, err = _run_bench(argv + ["--dry", "--steps", "2"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1
    ln = lines[0]
    n = int(argv[1])
    assert ln["n_gpus"] == n and ln["rccl_ranks"] == n and ln["collectives_ok"]
    assert ln["bucket_elems"] == bucket and ln["bucket_overlap"] is overlap and ln["sync_batchnorm_layers"] == sync_layers
    assert 0 < ln["bucket_early_elems"] < bucket
    assert 0.0 < ln["rank_ms_per_step"]["min"] <= ln["rank_ms_per_step"]["max"]


def test_bench_self_launch_fails_when_a_rank_dies():
    rc, lines, err = _run_bench(["--gpus", "3", "--dry", "--steps", "1"], GFS3D_BENCH_DRY_FAIL_RANK="1")
    assert rc != 0 and not lines
    assert "rank 1 exited with code 3" in err


def test_bench_under_an_external_launcher_keeps_working():
    """The torchrun form: WORLD_SIZE / RANK set from outside -> no second level of spawning; --gpus must agree."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(_free_port())
    procs = []
    for r in range(2):
        e = dict(os.environ, GFS3D_BENCH_BACKEND="gloo", WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry", "--steps", "1"], env=e,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    e = dict(os.environ, GFS3D_BENCH_BACKEND="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry"], env=e, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
