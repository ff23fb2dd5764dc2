#!/usr/bin/env python
"""Benchmark of the north-star hot path (BASELINE.json): DGCNN forward+backward
over synthetic B x 2048 x 9 blocks, k = 20 (config 2: B = 32 per GPU), blocks/s.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N --steps K --warmup W          # N > 1 without WORLD_SIZE in the environment: this process starts N
                                                         # fresh rank processes itself (before any GPU call) and forwards rank 0's line
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  GFS3D_BENCH_BACKEND=gloo python bench.py --gpus 8 --dry   # the launch + the step's collectives without kernels (no GPU needed)

One process per GPU; ranks own disjoint shards of the block batch (weak scaling:
32 blocks per GPU); the only data-path collective is one flat-bucket gradient
all-reduce (RCCL) per step.  A step = zero grads, train-mode forward (BatchNorm
batch statistics), backward, gradient all-reduce, Adam update.  Rank 0 prints ONE JSON line
(with `rccl_ranks` = the size of the process group and the device of every rank).

--workload c2 (default)  BASELINE configs[1]: DGCNN + EdgeConv fwd/bwd, B = 32 per GPU, fp32 accuracy
--workload c4            configs[3]: the whole CAPL training step (S3DIS shapes, 13 + 1 classes, K_w = 200), B = 32 per GPU
--workload c5            configs[4]: CAPL step with ScanNet shapes (21 + 1 classes, K_w = 180), B = 64 per GPU, reduced
                         precision (half forward / bf16 backward operands; --dtype bf16 for bf16 everywhere)

The timed region runs un-instrumented; the per-kernel times come from a second pass of the same K steps with the
library's hipEvent brackets on (events on the launch stream, from a pool: no allocation inside).  With the default
workload rank 0 afterwards measures the other kernels of the path on their BASELINE shapes (`extra`: word assignment,
get_pred, one Lloyd step of configs[2], attention, the configs[3] / configs[4] steps), each against its binding roof.
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

NPTS, KNN, CIN = 2048, 20, 9
PEAK_FP32_TFLOPS = 157.3       # MI355X dense fp32 (vector == fp32 MFMA) peak, MI355X_MICROARCH.md
PEAK_16BIT_TFLOPS = 2500.0     # dense bf16 / f16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0
KERNELS = {0: "knn_topk_kernel", 1: "ec_main_fwd_kernel", 2: "ec_main_bwd_pair_kernel", 3: "proto_kernels", 4: "ec_stats1_kernel",
           5: "kmeans_assign_kernel", 6: "pointwise_stack_fwd", 7: "pointwise_stack_bwd", 8: "attention_fwd", 9: "attention_bwd"}
WORKLOADS = {
    "c2": dict(model="dgcnn", B=32, dtype="f32", classes=0, base=0, Kw=0,
               text="configs[1]: S3DIS-shaped blocks, batch=32x2048x9 per GPU, k=20, DGCNN+EdgeConv fwd/bwd (train mode, BN batch stats) "
                    "+ grad all-reduce + Adam"),
    "c4": dict(model="capl", B=32, dtype="f32", classes=13, base=7, Kw=200,
               text="configs[3]: S3DIS S0 CAPL training step (DGCNN + attention + geometric words K_w=200 + prototype matching, 13+1 "
                    "classes), batch=32x2048x9 per GPU, k=20, fwd/bwd + grad all-reduce + Adam"),
    "c5": dict(model="capl", B=64, dtype="f16", classes=21, base=15, Kw=180,
               text="configs[4]: ScanNet S0 CAPL training step (21+1 classes, K_w=180), batch=64x2048x9 per GPU, k=20, reduced precision "
                    "(16-bit matrix-core operands, fp32 accumulate and storage; kNN fp32 bit-exact), fwd/bwd + grad all-reduce + Adam"),
}


def algorithmic_flops(kid, B, cfg=None):
    """Algorithmic flops of ONE step's launches of kernel `kid` (SURVEY.md 8(d) figures):
    kNN layer: 2 N^2 C + 3 N^2 per block; conv2 of an EdgeConv layer: N k 2*64*64 per block
    forward, 2x that (dgrad + wgrad) backward.  Returns (flops_per_step, launches_per_step)."""
    N, k = NPTS, KNN
    if kid == 0:
        return B * sum(2 * N * N * C + 3 * N * N for C in (CIN, 64, 64)), 3
    if kid == 1:
        return B * 3 * N * k * 2 * 64 * 64, 3
    if kid == 2:
        return B * 3 * N * k * 2 * (2 * 64 * 64), 3
    mlp = 2 * (192 * 512 + 512 * 256)          # point-wise stack 192 -> 512 -> 256, flops per point (forward)
    if cfg is not None and cfg.get("model") == "capl":      # + BaseLearner 256 -> 128 -> 64, q|k|v 256 -> 192, fusion (K_w + 192) -> 128
        mlp += 2 * (256 * 128 + 128 * 64) + 2 * 256 * 192 + 2 * (cfg["Kw"] + 192) * 128
    if kid == 6:
        return B * N * mlp, 1
    if kid == 7:
        return B * N * 2 * mlp, 1
    if kid == 8:
        return B * 4 * N * N * 64, 1           # S = q^T k and O = P v
    if kid == 9:
        return B * 10 * N * N * 64, 1          # five N x N x 64 products (S and dP recomputed for both sides counted once)
    return 0, 0


def _traffic_file(tag, workload):
    """Newest committed PMC summary of `workload`: profiles/<tag>*_<workload>_hbm_traffic.json, else the older per-bench files."""
    import glob
    for pat in (f"{tag}*_{workload}_hbm_traffic.json", f"{tag}*_hbm_traffic.json", "r05_d_hbm_traffic.json"):
        got = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        if got:
            return got[-1]
    return None


def hbm_traffic(kernel, tag, workload="c2"):
    """HBM bytes per launch of `kernel` from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, FETCH_SIZE doubled per the gfx950 note; scripts/collect_profiles.py) - a figure of the committed profile, not of
    this run."""
    path = _traffic_file(tag, workload)
    if path:
        with open(path) as f:
            d = json.load(f)
        for k, v in d.items():
            if k.startswith(kernel) and isinstance(v, dict):
                return v.get("traffic_bytes_per_launch"), "profiles/" + os.path.basename(path)
    return None, None


# kernels behind each profile bracket (ids of KERNELS), as regular expressions over the PMC summaries' kernel names.  Brackets 6-9
# cover GROUPS of kernels (a whole stack pass).  Row GEMM template arguments: <BN, PRO, EPI, OUT_CF, ...>; PRO 2 = backward
# prologue; EPI 3 = plain (channel-first output: the forward q|k|v projections; point-major: their backward)
BRACKET_KERNELS = {
    1: [r"^ec_main_fwd_kernel"],
    2: [r"^ec_main_bwd_pair_kernel", r"^ec_main_bwd_kernel"],
    6: [r"^pw_rowgemm6_kernel<\d+, [01], [01], ", r"^pw_rowgemm6_kernel<\d+, 0, 3, true", r"^pw_out_kernel", r"^pw_split_all_kernel", r"^pw_bn_finalize_kernel"],
    7: [r"^pw_bwd_in_kernel", r"^pw_wgrad6_kernel", r"^pw_wgrad_reduce_kernel", r"^pw_rowgemm6_kernel<\d+, 2, ",
        r"^pw_rowgemm6_kernel<\d+, 0, 3, false", r"^pw_bwd_coef_kernel"],
    8: [r"^attn_fwd_kernel"],
    9: [r"^attn_bwd_"],
}


def group_traffic(kid, workload, launches_per_step, tag="r06", reduced=False):
    """(HBM bytes per bracketed launch, source file, kernel names) of profile bracket `kid` in `workload`: the PMC bytes of every
    launch of the bracket's kernels during the profiled run / the run's steps / the bracket's launches per step.  Needs the
    workload's own PMC summary (its `_steps` field); None without one."""
    import re
    path = _traffic_file(tag, workload)
    pats = BRACKET_KERNELS.get(kid)
    if not path or not pats or (kid not in (1, 2) and not os.path.basename(path).endswith(f"_{workload}_hbm_traffic.json")):
        return None, None, None
    with open(path) as f:
        d = json.load(f)
    total, count, names = 0.0, 0, []
    for k, v in d.items():
        if isinstance(v, dict) and any(re.search(p, k) for p in pats):
            if kid in (1, 2) and k.rstrip().endswith("0>") == reduced:      # the EdgeConv kernels' last template argument: 0 = the fp32-accurate form
                continue
            total += v["traffic_bytes_per_launch"] * v["launches"]
            count += v["launches"]
            names.append(k)
    if not names:
        return None, None, None
    src = "profiles/" + os.path.basename(path)
    if kid in (1, 2):                             # one kernel launch per bracket: the launch-weighted mean
        return int(total / count), src, sorted(names)
    steps = d.get("_steps")
    if not steps:
        return None, None, None
    return int(total / steps / max(launches_per_step, 1e-9)), src, sorted(names)


def pipe_of(kid, dtype):
    """(name, peak TFLOP/s) of the pipe kernel `kid` EXECUTES on: the fp32-accurate GEMMs run as six bf16 piece products per
    fp32 product on the bf16 matrix cores (dense bf16 peak / 6), the reduced modes as one 16-bit product; kNN stays on the
    fp32 MFMA chain (bit-exact contract) behind a bf16 screen, priced at the fp32 peak only."""
    if kid == 0:
        return "fp32 MFMA chain + bf16 screen", None
    if dtype == "f32":
        return "bf16 matrix cores, 6 piece products per fp32 product (dense bf16 peak / 6)", PEAK_16BIT_TFLOPS / 6
    return "16-bit matrix cores, one product per fp32 product (dense 16-bit peak)", PEAK_16BIT_TFLOPS


def roofline_of(prof, steps, B, cfg, tag="r06", workload="c2"):
    """Roofline block of one profiled workload: the library kernel with the largest device time among those with a flop model,
    its role flops / average launch time against (a) the fp32 peak the results are priced at and (b) the pipe it executes on,
    and its HBM bytes per launch from the committed PMC passes."""
    cand = [k for k in prof if algorithmic_flops(k, B, cfg)[1] > 0]
    roof = {"bound": "mfma", "kernel": None, "achieved": None, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
    if not cand:
        return roof
    dom = max(cand, key=lambda k: prof[k][0])
    ms, n = prof[dom]
    # ids 0-2 bracket ONE launch each (3 per step: the EdgeConv layers); ids 6-9 bracket whole passes, several per step in the
    # CAPL model (backbone stack, BaseLearner, fusion, q|k|v): their flop model is per STEP, so is the time it is divided by
    flops_step, launches_step = algorithmic_flops(dom, B, cfg)
    avg_ms = ms / n
    achieved = flops_step / (ms / steps * 1e-3) / 1e12
    pipe_name, pipe_peak = pipe_of(dom, cfg["dtype"])
    name = KERNELS[dom] + ("<1>" if (dom == 2 and cfg["dtype"] != "f32") else "")
    # HBM bytes per bracketed launch from the committed PMC passes of THIS workload (profiles/<tag>_<workload>_hbm_traffic.json)
    traffic, src, traffic_kernels = group_traffic(dom, workload, n / steps, tag, reduced=cfg["dtype"] != "f32")
    # `frac` / `peak` follow the arithmetic the results are priced at: the fp32 peak for the fp32-accurate path; for the reduced
    # dtypes (configs[4]) the executed 16-bit pipe - `frac_fp32` and `frac_executed_pipe` stay as the explicit fields
    reduced = cfg["dtype"] != "f32" and pipe_peak
    if reduced:
        roof["peak"] = round(pipe_peak, 1)
    roof.update({"kernel": name, "achieved": round(achieved, 3), "frac": round(achieved / (pipe_peak if reduced else PEAK_FP32_TFLOPS), 4),
                 "frac_fp32": round(achieved / PEAK_FP32_TFLOPS, 4), "executed_pipe": pipe_name,
                 "peak_executed_pipe": round(pipe_peak, 1) if pipe_peak else None,
                 "frac_executed_pipe": round(achieved / pipe_peak, 4) if pipe_peak else None,
                 "traffic": traffic, "traffic_source": src, "traffic_kernels": traffic_kernels, "avg_launch_ms": round(avg_ms, 4), "launches": n,
                 "launches_per_step": round(n / steps, 2), "role_flops_per_step": int(flops_step)})
    roof["per_kernel"] = {}
    for kid in sorted(prof):
        fl, nl = algorithmic_flops(kid, B, cfg)
        ent = {"ms_per_step": round(prof[kid][0] / steps, 4)}
        if nl:
            tf = fl / (prof[kid][0] / steps * 1e-3) / 1e12
            _, pk = pipe_of(kid, cfg["dtype"])
            ent.update({"algorithmic_TFLOPps": round(tf, 2), "frac_fp32": round(tf / PEAK_FP32_TFLOPS, 4),
                        "frac_executed_pipe": round(tf / pk, 4) if pk else None})
        roof["per_kernel"][KERNELS[kid]] = ent
    return roof


def cpu_baseline(seconds_budget=60.0):
    """The oracle (torch-CPU restatement of model/dgcnn.py, kind='port') timed on this host's cores with BASELINE.md
    section 3's protocol: B = 1 eval forward (BASELINE configs[0]: 3 warm-ups, median of 10) and B = 32 train-mode
    forward + backward (the metric's own workload; 1 warm-up, median of up to 10 iterations within `seconds_budget`,
    at least 3).  `value` is the B = 32 forward + backward figure."""
    from oracle import ref_ops as R
    from model.dgcnn import DGCNN
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in DGCNN([[64, 64]] * 3, [512, 256], CIN, k=KNN, return_edgeconvs=True).state_dict().items()}
    prev_threads = torch.get_num_threads()
    cores = min(32, os.cpu_count() or 1)      # measured on the 2x64-core EPYC host: 8/16/32/64/128 threads -> 4.0/4.4/4.9/3.3/1.8 blocks/s
    torch.set_num_threads(cores)

    def median_time(fn, warm, lo, hi, budget):
        for _ in range(warm):
            fn()
        times, t_start = [], time.time()
        while len(times) < lo or (len(times) < hi and time.time() - t_start < budget):
            t0 = time.time()
            fn()
            times.append(time.time() - t0)
        ts = sorted(times)
        return ts[len(ts) // 2], len(ts), ts[0], ts[-1]

    x1 = R.synthetic_blocks(1, NPTS, seed=0)
    with torch.no_grad():
        dt1, n1, lo1, hi1 = median_time(lambda: R.dgcnn_forward(sd, x1, k=KNN, training=False), 3, 10, 10, 0.0)
    Bc = 32
    x = R.synthetic_blocks(Bc, NPTS, seed=0)

    def one():
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
        _, out = R.dgcnn_forward(p, x, k=KNN, training=True, new_stats={})
        out.mean().backward()
    dt, n, lo, hi = median_time(one, 1, 3, 10, seconds_budget)
    torch.set_num_threads(prev_threads)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(Bc / dt, 3), "value_min_max": [round(Bc / hi, 3), round(Bc / lo, 3)], "unit": "blocks/s", "cores": cores, "kind": "port",
            "cpu": model,
            "sample": f"median of {n} iterations (1 warm-up) of B={Bc} x {NPTS} x {CIN}, k={KNN}, DGCNN train-mode fwd+bwd (torch CPU oracle)",
            "b1_eval_forward": {"value": round(1.0 / dt1, 3), "value_min_max": [round(1.0 / hi1, 3), round(1.0 / lo1, 3)], "unit": "blocks/s",
                                "ms": round(dt1 * 1e3, 2), "ms_min_max": [round(lo1 * 1e3, 2), round(hi1 * 1e3, 2)],
                                "sample": f"BASELINE configs[0]: one 2048 x 9 block, DGCNN eval forward, median of {n1} after 3 warm-ups"}}


def capl_args():
    return types.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=CIN, dgcnn_k=KNN,
                                 base_widths=[128, 64], output_dim=64, eval_weight=1.2)


def build_step(cfg, dev, rank, world, sync_bn, gdist, force=False):
    """-> (step(), net, bucket, x): one training step of the workload on this rank's shard.  force: issue every collective of
    the N > 1 step although this is the only rank (--force-collectives: a 1-rank RCCL group)."""
    from gfs3d import ops
    from gfs3d.synthetic import synthetic_blocks    # input recipe; oracle/ is used by cpu_baseline() only
    torch.manual_seed(0)
    x = synthetic_blocks(cfg["B"], NPTS, seed=rank).to(dev)
    if os.environ.get("GFS3D_BENCH_MORTON") == "1":          # experiment: the same blocks, their points delivered in Morton order
        from gfs3d.morton import morton_order
        order, _ = morton_order(x[:, 0:3])
        x = torch.gather(x, 2, order[:, None, :].expand(-1, x.shape[1], -1)).contiguous()
    if cfg["model"] == "dgcnn":
        from model.dgcnn import DGCNN
        net = DGCNN([[64, 64]] * 3, [512, 256], CIN, k=KNN, return_edgeconvs=True).to(dev).train()
        early = list(net.conv.parameters())           # 88 % of the parameters, first in backward order
        y = None
    else:
        import contextlib
        import io
        from model.capl import mpti_net_Point_GeoAsWeight_v2
        gp = torch.randn(cfg["Kw"], 192, generator=torch.Generator().manual_seed(2)).to(dev)
        with contextlib.redirect_stdout(io.StringIO()):
            net = mpti_net_Point_GeoAsWeight_v2(classes=cfg["classes"], args=capl_args(), base_num=cfg["base"], gp=gp, energy=0.9)
        net = net.to(dev).train()
        edge = {id(p) for p in net.encoder.edge_convs.parameters()}
        early = [p for p in net.parameters() if id(p) not in edge]      # everything that is complete before the EdgeConv backward
        y = torch.randint(0, cfg["base"] + 1, (cfg["B"], NPTS), generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    multi = world > 1 or force
    if sync_bn and multi:
        net = gdist.enable_sync_batchnorm(net)
    # the early slice is exchanged under the EdgeConv backward - unless synchronised BatchNorm puts other collectives inside
    # backward (then every rank must issue them in one fixed order)
    bucket = gdist.FlatGradBucket(net.parameters(), early=early, overlap=not (sync_bn and multi))
    bucket.force_collectives = bool(force)
    if os.environ.get("GFS3D_BENCH_TORCH_ADAM") == "1":       # A/B switch (scripts/ab_env.sh): torch's own fused multi-tensor Adam
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    else:
        from gfs3d.optim import FusedAdam
        opt = FusedAdam(net.parameters(), lr=1e-3)       # torch.optim.Adam's update, all parameters in one launch (csrc/optim.hip)
    mode = cfg["dtype"]

    if cfg["model"] == "dgcnn":
        def step():
            bucket.zero()
            with ops.compute(mode):
                _, out = net(x)
                out.mean().backward()
            bucket.all_reduce_mean()
            opt.step()
    else:
        def step():
            bucket.zero()
            random.seed(0)                            # rank-identical fake-novel draw (train.py seeds every rank alike)
            with ops.compute(mode):
                _, loss = net(x, y=y)
                loss.backward()
            bucket.all_reduce_mean()
            opt.step()
    return step, net, bucket, x


def read_profile(L):
    prof = {}
    for kid in KERNELS:
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        L.gfs3d_profile_read(kid, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            prof[kid] = (ms.value, n.value)
    return prof


def timed_ms(fn, iters=50, warm=10, chunks=10):
    """Device time per call on the current stream (the stream the library launches on): SURVEY 8(d)'s protocol - `warm` untimed
    calls (a cold device ramps its clocks over the first few milliseconds), then `iters` calls in `chunks` event-bracketed groups;
    the MEDIAN of the groups' per-call times (one slow group - a clock step, another tenant's burst - does not move it)."""
    for _ in range(warm):
        fn()
    chunks = max(1, min(chunks, iters))
    per = max(1, iters // chunks)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    ev[0].record()
    for c in range(chunks):
        for _ in range(per):
            fn()
        ev[c + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[c].elapsed_time(ev[c + 1]) / per for c in range(chunks))
    return ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])


def c3_end_to_end(dev, n_blocks=1400, num_cnt=200):
    """configs[2] end to end (reference get_basis.py:162-222): M synthetic base-split blocks -> EdgeConv features of every
    point (resident) -> <= 300 000 points per foreground class -> k-means++ seeding -> Lloyd to sklearn's tolerance ->
    per-cluster means -> rank-truncated SVD (host numpy).  1400 blocks x 2048 points with labels in 0..7 give
    7 classes x ~358 000 points -> the full 7 x 300 000 x 192 k-means problem."""
    import types as _t
    from get_basis import DGCNNSeg, extract_geometric_words
    from gfs3d.synthetic import synthetic_blocks
    a = _t.SimpleNamespace(edgeconv_widths=[[64, 64]] * 3, dgcnn_mlp_widths=[512, 256], pc_in_dim=CIN, dgcnn_k=KNN)
    torch.manual_seed(0)
    seg = DGCNNSeg(a, num_classes=8).to(dev).eval()
    blocks = torch.cat([synthetic_blocks(100, NPTS, seed=100 + i) for i in range(n_blocks // 100)]).to(dev)   # resident: no PCIe in the flow
    labels = torch.randint(0, 8, (blocks.shape[0], NPTS), generator=torch.Generator().manual_seed(1)).to(dev)
    # warm-up (first-use costs) through the SAME code paths as the timed flow: 40 centroids (>= 33: the Lloyd loop builds its running
    # sums - ops.KMeansDelta, whose torch.aminmax is the first tensor-library reduction of the process: 166 ms of module loading in the
    # first process on a box, scripts/c3_first_call.py - and enters the run-ahead loop)
    extract_geometric_words(seg, blocks[:64], labels[:64], 8, 40, max_per_class=2000, max_iter=12)
    torch.cuda.synchronize()
    tm = {}
    t0 = time.perf_counter()
    words = extract_geometric_words(seg, blocks, labels, 8, num_cnt, max_per_class=300000, batch=64, seed=0, max_iter=300, timings=tm)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    host = tm["svd_host"] + tm["means_to_host"]
    it = tm["lloyd_iterations"]
    return {"workload": f"configs[2]: {blocks.shape[0]} synthetic blocks x {NPTS} pts, 7 foreground classes -> {tm['points']} x 192 points, "
                        f"{num_cnt} centroids, k-means++ then Lloyd to tol 1e-4 ({it} iterations), means, SVD",
            "total_ms": round(total * 1e3, 1),
            "stage_ms": {k: round(tm[k] * 1e3, 2) for k in ("features", "subsample", "seeding", "lloyd", "means_to_host", "svd_host")},
            "lloyd_ms_per_iteration": round(tm["lloyd"] * 1e3 / max(it + 1, 1), 3),
            "features_blocks_per_s": round(blocks.shape[0] / tm["features"], 1),
            "host_share": round(host / total, 4),
            "words_shape": list(words.shape),
            "note": "each stage closed by a device synchronisation; host = the 150 KB read-back of the means + numpy SVD (the only host "
                    "arithmetic); the Lloyd loop reads one scalar per iteration for its stopping test"}


def pipeline_throughput(dev, n_blocks=256, batch=32, iters=30):
    """blocks/s of the device input pipeline (gfs3d.pipeline.BlockSampler.batch: point draw, row gather, min-shift,
    augmentation, unit-box normalisation, label remap for a batch of 32 blocks = what 32 calls of the reference's
    sample_pointcloud do on the host, loader.py:39-151) on a resident synthetic dataset of S3DIS-like blocks (3 000 - 12 000
    points x 7 columns).  Both draw modes; the figure to hold against the training step's block rate."""
    from gfs3d import pipeline
    g = torch.Generator().manual_seed(11)
    blocks = []
    for _ in range(n_blocks):
        m = int(torch.randint(3000, 12001, (1,), generator=g))
        b = torch.rand(m, 7, generator=g)
        b[:, 3:6] *= 255.0
        b[:, 6] = torch.randint(0, 13, (m,), generator=g).float()
        blocks.append(b)
    smp = pipeline.BlockSampler(blocks, dev)
    cfg = {"scale": 0, "rot": 1, "mirror_prob": 0, "jitter": 1}                      # the reference's training augmentation (train.py:787-790)
    gen = torch.Generator(device=dev).manual_seed(0)
    ids = torch.randint(0, n_blocks, (iters + 5, batch), generator=g).tolist()
    out = {}
    for key, cls in (("random_sample", None), ("class_guided", 3)):
        def one(i):
            return smp.batch(ids[i], NPTS, sampled_classes=list(range(1, 8)), sampled_class=cls, pc_augm=True, pc_augm_config=cfg, generator=gen)
        for i in range(5):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5, 5 + iters):
            one(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        out[key] = {"ms_per_batch": round(dt * 1e3, 4), "blocks_per_s": round(batch / dt, 1)}
    return {"dataset": f"{n_blocks} resident blocks of 3 000 - 12 000 points x 7 columns, batches of {batch} x {NPTS} points, z-rotation + jitter",
            "blocks_per_s": out["random_sample"]["blocks_per_s"], **out,
            "note": "wall clock of back-to-back batches (host launch cost included); no device -> host read inside a batch"}


def extras(dev, L, gdist):
    """The other kernels of the path on their BASELINE shapes, each against its BINDING roof (word cosine,
    get_pred and the Lloyd step are HBM-bound, attention is bound by the fp32 matrix rate)."""
    from gfs3d import ops
    out = {}
    B, N = 32, NPTS
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        feat = torch.randn(B, 192, N, generator=g).to(dev)
        words = torch.randn(200, 192, generator=g).to(dev)
        for mode in ("f32", "f16"):
            with ops.compute(mode):
                ms = timed_ms(lambda: ops.word_assign_raw(feat, words, 10.0), 100)
            byts = 4.0 * B * N * (192 + 200 + 200 + 1)              # features in; softmax, one-hot and word id out
            out["word_assign_" + mode] = {"shape": "B=32x2048, C=192, K_w=200", "ms": round(ms, 4), "bound": "hbm",
                                          "algorithmic_GBps": round(byts / ms / 1e6, 1), "frac": round(byts / ms / 1e6 / PEAK_HBM_GBS, 4),
                                          "algorithmic_TFLOPps": round(2.0 * B * N * 192 * 200 / ms / 1e9, 2)}
        f2 = torch.randn(B, 128, N, generator=g).to(dev)
        pr = torch.randn(14, 128, generator=g).to(dev)
        ms = timed_ms(lambda: ops.proto_cosine_raw(f2, pr, 10.0), 100)
        byts = 4.0 * B * N * (128 + 14)
        out["get_pred"] = {"shape": "B=32x2048, C=128, K=13+1", "ms": round(ms, 4), "bound": "hbm",
                           "algorithmic_GBps": round(byts / ms / 1e6, 1), "frac": round(byts / ms / 1e6 / PEAK_HBM_GBS, 4)}
        # the fused prototype head of the training step (capl.py:194-242), forward + backward, S3DIS shapes
    f3 = (torch.randn(B, 128, N, generator=g).to(dev)).requires_grad_(True)
    mp = torch.randn(13, 128, generator=g).to(dev).requires_grad_(True)
    bgp = torch.randn(1, 128, generator=g).to(dev).requires_grad_(True)
    yh = torch.randint(0, 8, (B, N), generator=g).to(dev)

    def head():
        f3.grad = mp.grad = bgp.grad = None
        ops.capl_train_head(f3, mp, bgp, yh, [2, 4, 7], 7, 255)[1].backward()
    ms = timed_ms(head, 100)
    # the same forward + backward replayed as ONE hipGraph: the figure a host that cannot queue 11 launches in 0.13 ms does not move
    graph_ms, graph_note = None, None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                head()
        torch.cuda.current_stream().wait_stream(side)
        gph = torch.cuda.CUDAGraph()
        f3.grad = mp.grad = bgp.grad = None
        with torch.cuda.graph(gph):
            ops.capl_train_head(f3, mp, bgp, yh, [2, 4, 7], 7, 255)[1].backward()
        graph_ms = round(timed_ms(gph.replay, 100), 4)
        del gph
    except Exception as exc:                                          # a capture failure must not take the bench line with it
        graph_note = f"graph capture failed: {type(exc).__name__}: {exc}"[:200]
    byts = 4.0 * B * N * 128 * 3                                     # features read by the forward and by the backward, gradient written
    head_flops = 2.0 * B * N * 128 * (14 * 11 + 3 * 14)             # 11 class-table products of 14 x 128 per point + the three point contractions
    out["capl_train_head"] = {"shape": "B=32x2048, C=128, 13+1 classes, 3 fake-novel classes; forward + backward", "ms": round(ms, 4),
                              "graph_replay_ms": graph_ms, **({"graph_note": graph_note} if graph_note else {}),
                              "bound": "hbm (its bytes are the floor; the four point passes run on the fp32 matrix cores, v_mfma_f32_16x16x4_f32, "
                                       "at ~2x their matrix floor; the 11 launches are host-latency bound)",
                              "algorithmic_GBps": round(byts / ms / 1e6, 1), "frac": round(byts / ms / 1e6 / PEAK_HBM_GBS, 4),
                              "executed_TFLOPps_fp32": round(head_flops / ms / 1e9, 2),
                              "note": "generate_fake_proto + 3 x get_pred + post_refine_proto_v2 + 2 x CrossEntropyLoss + arg-max as 11 "
                                      "launches (csrc/head.hip); the tensor-library form it replaces took ~1.45 ms in ~200 launches"}
    del f3, mp, bgp, yh
    with torch.no_grad():
        # configs[2]: one Lloyd step over the full S3DIS base split (7 classes x 300 000 points, 192-d, 200 centroids)
        M = 7 * 300000
        pts = torch.randn(M, 192, generator=g).to(dev)
        cent = pts[torch.randperm(M, generator=g)[:200].to(dev)].contiguous()
        byts = 4.0 * M * (192 + 1)                                   # every point read once, its label written
        for key, fast, note in (("lloyd_step", True, "the step of the Lloyd loop (get_basis.device_kmeans): SCREENED assignment (hi + mid bf16 operands, three "
                                 "products, error bound, the whole operand image of the centroids resident in LDS; undecided points re-scored at fp32 accuracy from a list: same labels / sums / counts as the exact "
                                 "form) + centroid sums as sorted segment sums over the rows (the Lloyd LOOP itself replaces that second pass by delta "
                                 "sums over the moved rows once < 3 % of the labels change: gfs3d_kmeans_delta_fix)"),
                                ("lloyd_step_exact", False, "every point scored at fp32 accuracy (bf16x6 GEMM): the final assignment, whose inertia is reported")):
            ms = timed_ms(lambda: ops.kmeans_assign(pts, cent, fast=fast), 50)
            fl = 2.0 * M * 192 * 200
            out[key] = {"shape": "M=2.1e6, C=192, K=200 (configs[2])", "ms": round(ms, 3),
                        "bound": "mfma (bf16x6: the scoring GEMM is 2 K = 400 flop per 4 B read = 100 flop/B, above the fp32 balance; "
                                 "the screened form executes 3 of the 6 piece products)",
                        "algorithmic_TFLOPps": round(fl / ms / 1e9, 2),
                        "frac_fp32": round(fl / ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                        "frac_executed_pipe": round(fl / ms / 1e9 / (PEAK_16BIT_TFLOPS / (3 if fast else 6)), 4),
                        "algorithmic_GBps": round(byts / ms / 1e6, 1), "frac_hbm": round(byts / ms / 1e6 / PEAK_HBM_GBS, 4), "note": note}
        del pts, cent
        torch.cuda.empty_cache()
        out["c3_end_to_end"] = c3_end_to_end(dev)
        out["pipeline"] = pipeline_throughput(dev)
        out["pipeline_blocks_per_s"] = out["pipeline"]["blocks_per_s"]
    q, k_, v = (torch.randn(B, 64, N, generator=g).to(dev).requires_grad_(True) for _ in range(3))
    cot = torch.randn(B, 64, N, generator=g).to(dev)

    def att():
        q.grad = k_.grad = v.grad = None
        ops.attention(q, k_, v, 0.125, 0.1, 1234).backward(cot)
    read_profile(L)
    for _ in range(3):
        att()
    torch.cuda.synchronize()
    L.gfs3d_profile_enable(1)
    for _ in range(10):
        att()
    torch.cuda.synchronize()
    L.gfs3d_profile_enable(0)
    prof = read_profile(L)
    for kid, key in ((8, "attention_fwd"), (9, "attention_bwd")):
        if kid in prof:
            ms = prof[kid][0] / prof[kid][1]
            fl = algorithmic_flops(kid, B)[0]
            # fp32 results on the bf16 matrix cores (6 piece products per fp32 product): roof = dense bf16 peak / 6
            out[key] = {"shape": "B=32x2048, d=64, dropout 0.1", "ms": round(ms, 4), "bound": "mfma (bf16x6: dense bf16 peak / 6)",
                        "algorithmic_TFLOPps": round(fl / ms / 1e9, 2), "peak_executed_pipe": round(PEAK_16BIT_TFLOPS / 6, 1),
                        "frac_fp32": round(fl / ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                        "frac_executed_pipe": round(fl / ms / 1e9 / (PEAK_16BIT_TFLOPS / 6), 4)}
    del q, k_, v, cot
    for name in ("c4", "c5"):
        cfg = WORKLOADS[name]
        step, net, bucket, x = build_step(cfg, dev, 0, 1, False, gdist)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        chunk_ms = []
        for _ in range(6):                        # wall clock like the headline: 6 groups of 5 steps, the median group
            t0 = time.perf_counter()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            chunk_ms.append((time.perf_counter() - t0) / 5 * 1e3)
        chunk_ms.sort()
        dt = 0.5 * (chunk_ms[2] + chunk_ms[3]) * 1e-3
        n_prof = 10
        L.gfs3d_profile_enable(1)
        for _ in range(n_prof):
            step()
        torch.cuda.synchronize()
        L.gfs3d_profile_enable(0)
        prof = read_profile(L)
        out["step_" + name] = {"workload": cfg["text"], "dtype": cfg["dtype"], "ms_per_step": round(dt * 1e3, 3),
                               "ms_per_step_min_max": [round(chunk_ms[0], 3), round(chunk_ms[-1], 3)],
                               "blocks_per_s": round(cfg["B"] / dt, 1),
                               "library_kernel_ms_per_step": {KERNELS[k]: round(v[0] / n_prof, 4) for k, v in prof.items()},
                               "roofline": roofline_of(prof, n_prof, cfg["B"], cfg, workload=name)}
        if name == "c4":       # the testing path of the reference (train.py test loop): capl.forward(eval_model=True), fp32 results
            net.eval()
            ncls = cfg["classes"]
            gened = torch.randn(ncls, 128, generator=torch.Generator().manual_seed(3)).to(dev)
            bc = (torch.rand(cfg["base"], cfg["Kw"], generator=torch.Generator().manual_seed(4)) > 0.5).float().to(dev)
            nc = (torch.rand(ncls - cfg["base"], cfg["Kw"], generator=torch.Generator().manual_seed(5)) > 0.5).float().to(dev)
            yy = torch.randint(0, ncls, (cfg["B"], NPTS), generator=torch.Generator().manual_seed(6)).to(dev)
            with torch.no_grad():
                ms = timed_ms(lambda: net(x, y=yy, gened_proto=gened, eval_model=True, base_class_coding=bc, novel_class_coding=nc), 50)
            out["capl_eval_forward"] = {"workload": "configs[3] shapes, capl.forward(eval_model=True): backbone + heads + prototype "
                                        "refinement + geometric-word re-weighting, B=32x2048", "ms": round(ms, 4),
                                        "blocks_per_s": round(cfg["B"] / (ms * 1e-3), 1)}
        del step, net, bucket, x
        torch.cuda.empty_cache()
    return out


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without an external launcher: start N fresh child interpreters (one rank per GPU) with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, forward rank 0's JSON line, exit non-zero if any child did.  The parent
    never touches the GPU (no HIP call, not even torch.cuda.is_available()): the children are plain new processes, not
    an exec of a process that initialised the device."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, GFS3D_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    deadline = time.time() + float(os.environ.get("GFS3D_BENCH_TIMEOUT", "1500"))
    rc, out0 = 0, ""
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                try:
                    if r == 0:
                        out0 = procs[0].communicate(timeout=0.2)[0] or ""
                    else:
                        procs[r].wait(timeout=0.2)
                except subprocess.TimeoutExpired:
                    continue
                live.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode
                    print(f"bench.py: rank {r} exited with code {procs[r].returncode}", file=sys.stderr)
            if (rc or time.time() > deadline) and live:
                rc = rc or 124
                break
    finally:
        for r in live:                   # a failed or overdue job: end exactly the children started here
            procs[r].kill()
        for r in live:
            procs[r].wait()
    lines = []
    for ln in out0.splitlines():                 # rank 0's result line (library chatter on its stdout is not forwarded)
        if ln.startswith("{"):
            try:
                if "n_gpus" in json.loads(ln):
                    lines.append(ln)
            except ValueError:
                pass
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} result lines", file=sys.stderr)
        rc = 1
    for ln in lines[-1:]:
        print(ln, flush=True)
    return rc


def dry_run(args, world, rank, local):
    """--dry: the N-rank flow without kernels (runs on a box without a GPU): build the model's flat gradient bucket, run
    the step's collectives on it (early slice + remainder, mean), the barrier and max-over-ranks of the timed region, and
    print the line's launch-related fields.  Exercised by tests/test_dist_cpu.py with GFS3D_BENCH_BACKEND=gloo --gpus 8."""
    import torch.distributed as dist
    from gfs3d import dist as gdist
    backend = os.environ.get("GFS3D_BENCH_BACKEND", "nccl")
    on_gpu = backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("GFS3D_BENCH_DRY_FAIL_RANK") == str(rank):      # tests: a rank that dies must fail the whole job
        os._exit(3)
    cfg = WORKLOADS[args.workload]
    sync_bn = bool(args.sync_bn and world > 1)
    torch.manual_seed(0)
    if cfg["model"] == "dgcnn":
        from model.dgcnn import DGCNN
        net = DGCNN([[64, 64]] * 3, [512, 256], CIN, k=KNN, return_edgeconvs=True).to(dev)
        early = list(net.conv.parameters())
    else:                                         # the CAPL model of configs[3] / configs[4], as build_step() makes it
        import contextlib
        import io
        from model.capl import mpti_net_Point_GeoAsWeight_v2
        gp = torch.randn(cfg["Kw"], 192, generator=torch.Generator().manual_seed(2)).to(dev)
        with contextlib.redirect_stdout(io.StringIO()):
            net = mpti_net_Point_GeoAsWeight_v2(classes=cfg["classes"], args=capl_args(), base_num=cfg["base"], gp=gp, energy=0.9).to(dev)
        edge = {id(p) for p in net.encoder.edge_convs.parameters()}
        early = [p for p in net.parameters() if id(p) not in edge]
    n_sync = 0
    if sync_bn:                                   # what `--sync-bn` turns on in build_step(): SyncBatchNorm heads + BNSync on the EdgeConv phases
        net = gdist.enable_sync_batchnorm(net)
        n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in net.modules())
    bucket = gdist.FlatGradBucket(net.parameters(), early=early, overlap=not sync_bn)
    encoder = net if cfg["model"] == "dgcnn" else net.encoder
    ok = True
    t0 = time.perf_counter()
    for it in range(args.steps):
        bucket.zero()
        if sync_bn:
            # the order of one synchronised step: two fp64 phase sums per EdgeConv layer forward, two backward (gfs3d.ops.BNSync:
            # the all-reduces the phased K2 entry points wait for), all on the bucket's communicator - hence overlap=False above
            for _ in range(4 * len(encoder.edge_convs)):
                acc = torch.full((128,), float(rank + 1), dtype=torch.float64, device=dev)
                encoder.bn_sync.all_reduce(acc)
                ok = ok and bool((acc == world * (world + 1) / 2.0).all())
        if cfg["model"] == "capl":
            # the two data-path collectives of capl.forward under block-sharded DP, through the model's own code: label
            # presence (MAX) and the fake-novel (feature sum, count) all-reduce of generate_fake_proto (capl.py:364-411)
            import random
            g = torch.Generator().manual_seed(100 * it + rank)
            feat = torch.randn(2, net.main_proto.shape[1], 64, generator=g).to(dev)
            y = torch.randint(rank % 3, cfg["base"] + 1, (2, 64), generator=g).to(dev)
            random.seed(0)
            proto, chosen = net.generate_fake_proto(feat, y, net.main_proto.detach())
            got = [None] * world
            if world > 1:
                dist.all_gather_object(got, (sorted(int(c) for c in chosen), float(proto.double().sum())))
                ok = ok and all(g_[0] == got[0][0] and abs(g_[1] - got[0][1]) < 1e-4 for g_ in got)
        for p in net.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        bucket.all_reduce_mean()
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    want = (world + 1) / 2.0
    ok = ok and (world == 1 or bool(torch.allclose(bucket.flat, torch.full_like(bucket.flat, want))))
    devices = rank_devices(dev, world)
    per_rank = rank_values(mine / max(args.steps, 1) * 1e3, world)
    if rank == 0:
        print(json.dumps({"dry": True, "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": backend,
                          "workload": args.workload, "sync_bn": sync_bn, "sync_batchnorm_layers": n_sync,
                          "devices": devices, "steps": args.steps, "bucket_elems": int(bucket.flat.numel()),
                          "bucket_early_elems": int(bucket.n_early), "bucket_overlap": bool(bucket.overlap),
                          "collectives_ok": ok, "elapsed_s": round(elapsed, 4),
                          "rank_ms_per_step": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def rank_values(value, world):
    """One float per rank, gathered to all ranks (rank 0 prints min / max: a straggler shows)."""
    import torch.distributed as dist
    if world == 1:
        return [float(value)]
    got = [None] * world
    dist.all_gather_object(got, float(value))
    return got


def rank_devices(dev, world):
    """Device of every rank, gathered to all ranks (evidence that N ranks drove N cards)."""
    import torch.distributed as dist
    mine = f"{dev.type}:{dev.index}" if dev.type == "cuda" else "cpu"
    if world == 1:
        return [mine]
    got = [None] * world
    dist.all_gather_object(got, mine)
    return got


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry", action="store_true", help="launch + collectives only, no kernels (works without a GPU under "
                                                       "GFS3D_BENCH_BACKEND=gloo)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2")
    ap.add_argument("--dtype", choices=["f32", "f16", "bf16"], default=None, help="override the workload's arithmetic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements of the other kernels (rank 0, N = 1)")
    ap.add_argument("--sync-bn", action="store_true", help="whole-batch BatchNorm statistics across ranks (N>1)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: run the step's collectives anyway, through a 1-rank RCCL group (the async early slice, the "
                         "remainder all-reduce and, with --sync-bn, the BatchNorm phase all-reduces go through real RCCL streams "
                         "inside the timed step) - first-contact rehearsal for the multi-GPU run; not the default measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:       # no external launcher: this process becomes one (no GPU call before this)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry:
        raise SystemExit(dry_run(args, world, rank, local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # GFS3D_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("GFS3D_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    forced = bool(args.force_collectives and world == 1)
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gfs3d import _lib, dist as gdist
    L = _lib.lib()
    cfg = dict(WORKLOADS[args.workload])
    if args.dtype:
        cfg["dtype"] = args.dtype
    B = cfg["B"]
    step, net, bucket, x = build_step(cfg, dev, rank, world, args.sync_bn, gdist, force=forced)
    probe = None
    if forced:
        # latency of the step's two gradient messages and of one BatchNorm phase message on this rank's RCCL stream (one rank:
        # no wire - launch + kernel cost, the floor every N pays; DESIGN.md section 5's prediction table starts from it)
        def _ar(t):
            return lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t64 = torch.zeros(128, dtype=torch.float64, device=dev)
        probe = {"early_slice_us": round(timed_ms(_ar(bucket.flat[:max(bucket.n_early, 1)])) * 1e3, 2),
                 "remainder_us": round(timed_ms(_ar(bucket.flat[bucket.n_early:] if bucket.n_early < bucket.flat.numel() else bucket.flat)) * 1e3, 2),
                 "bn_phase_128_f64_us": round(timed_ms(_ar(t64)) * 1e3, 2),
                 "early_bytes": int(4 * bucket.n_early), "remainder_bytes": int(4 * (bucket.flat.numel() - bucket.n_early))}

    if world > 1 or forced:          # communicator set-up and the first-use costs of both message sizes stay out of the timed region for any W
        dist.all_reduce(torch.zeros_like(bucket.flat))
        dist.all_reduce(torch.zeros(max(bucket.n_early, 1), device=dev))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = rank_values(mine / args.steps * 1e3, world)          # each rank's own K steps, before it waited for the others
    elapsed = gdist.max_over_ranks(elapsed, dev)
    devices = rank_devices(dev, world)

    # second pass, outside the timed region: the same K steps with the library's hipEvent brackets on (all ranks: the
    # steps hold collectives)
    L.gfs3d_profile_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    L.gfs3d_profile_enable(0)
    prof = read_profile(L)
    if world > 1:
        dist.barrier()

    # north-star side figure (BASELINE.json: "kNN + EdgeConv forward at B=32 x 2048"): the three kNN + EdgeConv
    # blocks alone, eval mode (running statistics), after and outside the timed region
    fwd_ms = c1_ms = None
    if rank == 0 and cfg["model"] == "dgcnn":
        from gfs3d import ops
        net.eval()
        with torch.no_grad(), ops.compute(cfg["dtype"]):
            fwd_ms = timed_ms(lambda: net.edgeconv_features(x), 50)
            c1_ms = timed_ms(lambda: net(x[:1]), 50)          # BASELINE configs[0] on the device: one 2048 x 9 block, DGCNN eval forward
        net.train()

    if rank == 0:
        roof = roofline_of(prof, args.steps, B, cfg, workload=args.workload)
        if roof["kernel"] and roof["kernel"].startswith("ec_main_bwd_pair_kernel"):
            roof["traffic_note"] = ("ec_main_bwd_pair_kernel: ~67 MB read (P', Q', neighbour ids, winners, gradients) + ~357 MB of plain 256-B row "
                                    "stores, 335 MB of them the per-edge dy1 rows in CSR order that ec_bwd_segsum_kernel reads back once, last-written "
                                    "clouds first (the fixed-order, atomic-free form of the scatter_add_ in get_edge_feature's backward)")
        roof["note"] = ("achieved = algorithmic flops of the kernel's role / its average launch time (second, instrumented pass, hipEvents on the "
                        "launch stream). `frac` = `frac_fp32`: against the fp32 peak the results are priced at (they are fp32-accurate); "
                        "`frac_executed_pipe`: against the pipe the kernel runs on (`executed_pipe`).  kNN stays on the bit-exact fp32 MFMA chain")
        if 0 in prof:       # "kNN HBM GB/s" of the metric: algorithmic bytes 4CN + 8Nk per block and layer
            kb = B * sum(4 * C * NPTS + 8 * NPTS * KNN for C in (CIN, 64, 64))
            kms = prof[0][0] / args.steps
            kfl = algorithmic_flops(0, B)[0] / (kms * 1e-3) / 1e12
            roof["knn"] = {"ms_per_step": round(kms, 4), "algorithmic_GBps": round(kb / (kms * 1e-3) / 1e9, 2),
                           "algorithmic_TFLOPps": round(kfl, 2), "frac_of_fp32_peak": round(kfl / PEAK_FP32_TFLOPS, 4)}
            # "kNN HBM GB/s" as MEASURED by the committed PMC passes: FETCH (x2) + WRITE bytes of the kNN kernels of one
            # step (C = 9 layer: <20, 1, QW> screen + <20, 16> exact + knn_prep_kernel (rows, image, norms); two C = 64 layers:
            # knn_img_kernel + <20, 4, QW> + <20, 64>)
            per_step = {"knn_screen_kernel<20, 1,": 1, "knn_exact_kernel<20, 16>": 1, "knn_prep_kernel": 1, "knn_img_kernel": 2,
                        "knn_screen_kernel<20, 4,": 2, "knn_exact_kernel<20, 64>": 2}
            tb, src = 0, None
            for name, n in per_step.items():
                t_, s_ = hbm_traffic(name, "r06")
                if t_:
                    tb, src = tb + n * t_, s_
            if tb:
                roof["knn"].update({"measured_hbm_bytes_per_step": tb, "measured_hbm_GBps": round(tb / (kms * 1e-3) / 1e9, 1),
                                    "measured_source": src})
        if fwd_ms:
            # SURVEY 8(d) per block: kNN 1.19 GF + EdgeConv fwd 2.44 GF (conv1 counted on the N k edges); compulsory bytes 2.10 MB + 3.80 MB.
            # The kernels run conv1 on the N points (P / Q algebra): executed-equivalent 1.19 + 3 x 0.336 (conv2) + 0.05 (projections) GF.
            # 565-640 flop/B: the fp32 roof binds, not HBM (the north star's "HBM roofline" cannot: 189 MB is 24 us at 8 TB/s)
            gf, mb = B * (1.19 + 2.44), B * (2.10 + 3.80)
            gf_exec = B * (1.19 + 3 * NPTS * KNN * 2 * 64 * 64 / 1e9 + 2 * NPTS * 64 * (2 * CIN + 4 * 64) / 1e9)
            roof["fwd_knn_edgeconv_eval"] = {"ms": round(fwd_ms, 4), "blocks_per_s": round(B / (fwd_ms * 1e-3), 1),
                                             "executed_equivalent_TFLOPps": round(gf_exec / fwd_ms, 2),
                                             "frac_fp32": round(gf_exec / fwd_ms / PEAK_FP32_TFLOPS, 4),
                                             "survey_priced_TFLOPps": round(gf / fwd_ms, 2),
                                             "survey_priced_note": "SURVEY 8(d) prices conv1 on the N k edges and all N^2 distances; the kernels run conv1 on the "
                                                                   "N points and screen 98 % of the distance work away - a throughput figure on the survey's "
                                                                   "flop count, NOT a roofline fraction (it can exceed the peak)",
                                             "algorithmic_GBps": round(mb / fwd_ms, 1)}
        if c1_ms:
            roof["c1_single_block_eval_forward"] = {"ms": round(c1_ms, 4), "blocks_per_s": round(1e3 / c1_ms, 1),
                                                    "note": "BASELINE configs[0] on the device (latency-bound: 32 of 256 CUs have work); "
                                                            "the CPU oracle's figure is cpu_baseline.b1_eval_forward"}
        ms_step = elapsed / args.steps * 1e3
        metric = "point-cloud blocks/sec (2048 pts, k=20) DGCNN fwd+bwd" if cfg["model"] == "dgcnn" else \
                 "point-cloud blocks/sec (2048 pts, k=20) CAPL training step"
        line = {
            "metric": metric, "value": round(world * B / (elapsed / args.steps), 2),
            "unit": "blocks/s", "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "devices": devices,
            "backend": backend if (world > 1 or forced) else None, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "rank_ms_per_step": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": cfg["text"], "blocks_per_gpu": B, "points": NPTS, "k": KNN,
                       "parallelism": f"block-sharded dp{world}" + ("+syncbn" if args.sync_bn and (world > 1 or forced) else "")},
            "roofline": roof,
        }
        if forced:
            line["forced_collectives"] = {"early_slice_launches": int(bucket.early_launches), "bucket_overlap": bool(bucket.overlap),
                                          "sync_bn": bool(args.sync_bn), "rccl_probe": probe,
                                          "note": "N = 1 with every collective of the N > 1 step issued on a 1-rank RCCL group inside the timed "
                                                  "region (rehearsal; the default N = 1 line issues none)"}
        if world == 1 and not args.no_extras and not forced and args.workload == "c2":
            del step, net, bucket
            torch.cuda.empty_cache()
            line["extra"] = extras(dev, L, gdist)
        if world == 1 and not args.no_cpu_baseline and not forced:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1 or forced:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
