#!/usr/bin/env python
"""Benchmark of the north-star hot path (BASELINE.json): DGCNN forward+backward
over synthetic B x 2048 x 9 blocks, k = 20 (config 2: B = 32 per GPU), blocks/s.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; ranks own disjoint shards of the block batch (weak scaling:
32 blocks per GPU); the only data-path collective is one flat-bucket gradient
all-reduce (RCCL) per step.  A step = zero grads, DGCNN train-mode forward
(BatchNorm batch statistics), backward, gradient all-reduce, Adam update.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

B_PER_GPU, NPTS, KNN, CIN = 32, 2048, 20, 9
PEAK_FP32_TFLOPS = 157.3       # MI355X dense fp32 (vector == matrix) peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
KERNELS = {0: "knn_topk_kernel", 1: "ec_main_fwd_kernel", 2: "ec_main_bwd_kernel", 3: "proto_mfma_kernel",
           4: "ec_stats1_kernel", 5: "kmeans_assign_kernel", 6: "pointwise_stack_fwd", 7: "pointwise_stack_bwd"}


def algorithmic_flops(kid, B):
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
    if kid == 6:
        return B * N * mlp, 1
    if kid == 7:
        return B * N * 2 * mlp, 1
    return 0, 0


def hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile (profiles/r01_e_hbm_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE doubled per the gfx950 note;
    scripts/collect_profiles.py)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_e_hbm_traffic.json")) as f:
            return json.load(f).get(kernel, {}).get("traffic_bytes_per_launch")
    except OSError:
        return None


def cpu_baseline(seconds_budget=25.0):
    """The oracle (torch-CPU restatement of model/dgcnn.py, kind='port') timed on this
    host's cores on a bounded sample: B=8 blocks of the same workload, fwd+bwd."""
    from oracle import ref_ops as R
    from model.dgcnn import DGCNN
    torch.manual_seed(0)
    sd = {k: v.clone() for k, v in DGCNN([[64, 64]] * 3, [512, 256], CIN, k=KNN, return_edgeconvs=True).state_dict().items()}
    Bc = 8
    x = R.synthetic_blocks(Bc, NPTS, seed=0)
    prev_threads = torch.get_num_threads()
    cores = min(32, os.cpu_count() or 1)      # measured on the 2x64-core EPYC host: 8/16/32/64/128 threads -> 4.0/4.4/4.9/3.3/1.8 blocks/s
    torch.set_num_threads(cores)

    def one():
        p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
        _, out = R.dgcnn_forward(p, x, k=KNN, training=True, new_stats={})
        out.mean().backward()
    one()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < seconds_budget * 0.6 and n < 10):
        one()
        n += 1
    dt = (time.time() - t0) / n
    torch.set_num_threads(prev_threads)
    return {"value": round(Bc / dt, 3), "unit": "blocks/s", "cores": cores, "kind": "port",
            "sample": f"{n} iterations of B={Bc} x {NPTS} x {CIN}, k={KNN}, DGCNN fwd+bwd (torch CPU oracle)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true", help="whole-batch BatchNorm statistics across ranks (N>1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # GFS3D_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("GFS3D_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from gfs3d import _lib, dist as gdist
    from model.dgcnn import DGCNN
    from gfs3d.synthetic import synthetic_blocks    # input recipe; oracle/ is used by cpu_baseline() only
    L = _lib.lib()

    torch.manual_seed(0)
    net = DGCNN([[64, 64]] * 3, [512, 256], CIN, k=KNN, return_edgeconvs=True).to(dev).train()
    if args.sync_bn and world > 1:
        net = gdist.enable_sync_batchnorm(net)
    # conv: 88 % of the parameters, first in backward: its slice is exchanged under the EdgeConv backward - unless synchronised
    # BatchNorm puts other collectives inside backward (then every rank must issue them in one fixed order)
    bucket = gdist.FlatGradBucket(net.parameters(), early=net.conv.parameters(), overlap=not (args.sync_bn and world > 1))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
    x = synthetic_blocks(B_PER_GPU, NPTS, seed=rank).to(dev)

    def step():
        bucket.zero()
        _, out = net(x)
        out.mean().backward()
        bucket.all_reduce_mean()
        opt.step()

    if world > 1:          # communicator set-up and the first-use costs of both message sizes stay out of the timed region for any W
        dist.all_reduce(torch.zeros_like(bucket.flat))
        dist.all_reduce(torch.zeros(max(bucket.n_early, 1), device=dev))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    L.gfs3d_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    L.gfs3d_profile_enable(0)
    elapsed = gdist.max_over_ranks(elapsed, dev)

    # north-star side figure (BASELINE.json: "kNN + EdgeConv forward at B=32 x 2048"): the three kNN + EdgeConv
    # blocks alone, eval mode (running statistics), after and outside the timed region
    fwd_ms = None
    if rank == 0:
        net.eval()
        with torch.no_grad():
            for _ in range(3):
                net.edgeconv_features(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                net.edgeconv_features(x)
            e1.record()
            torch.cuda.synchronize()
            fwd_ms = e0.elapsed_time(e1) / 20
        net.train()

    # per-kernel device time from the in-library hipEvent brackets (same stream as the launches)
    import ctypes
    prof = {}
    for kid, name in KERNELS.items():
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        L.gfs3d_profile_read(kid, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            prof[kid] = (ms.value, n.value)
    if rank == 0:
        dom = max((k for k in prof if k in (0, 1, 2)), key=lambda k: prof[k][0])
        ms, n = prof[dom]
        flops_step, launches_step = algorithmic_flops(dom, B_PER_GPU)
        avg_ms = ms / n
        achieved = (flops_step / launches_step) / (avg_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": KERNELS[dom], "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": hbm_traffic(KERNELS[dom]),
                "avg_launch_ms": round(avg_ms, 4), "launches": n,
                "note": "achieved = algorithmic fp32 flops / time against the fp32 (vector == fp32-MFMA) peak; the EdgeConv and "
                        "point-wise GEMMs run on the bf16 matrix cores at fp32 accuracy (exact 3-way bf16 split, 6 piece "
                        "products per fp32 product); kNN stays on the bit-exact fp32 MFMA chain",
                "per_kernel_ms_per_step": {KERNELS[k]: round(v[0] / args.steps, 4) for k, v in prof.items()}}
        # the point-wise stack runs on the bf16 matrix cores at fp32 accuracy (6 bf16 products per fp32 product):
        # algorithmic fp32 TFLOP/s of its two launch groups (all kernels of a pass), against the same fp32 peak
        for kid in (6, 7):
            if kid in prof:
                fl, _ = algorithmic_flops(kid, B_PER_GPU)
                roof[KERNELS[kid]] = {"ms_per_step": round(prof[kid][0] / args.steps, 4),
                                      "algorithmic_TFLOPps": round(fl / (prof[kid][0] / args.steps * 1e-3) / 1e12, 2)}
        if 0 in prof:       # "kNN HBM GB/s" of the metric: algorithmic bytes 4CN + 8Nk per block and layer
            kb = B_PER_GPU * sum(4 * C * NPTS + 8 * NPTS * KNN for C in (CIN, 64, 64))
            kms = prof[0][0] / args.steps
            roof["knn"] = {"ms_per_step": round(kms, 4), "algorithmic_GBps": round(kb / (kms * 1e-3) / 1e9, 2),
                           "algorithmic_TFLOPps": round(algorithmic_flops(0, B_PER_GPU)[0] / (kms * 1e-3) / 1e12, 2)}
        if fwd_ms:
            # SURVEY 8(d) per block: kNN 1.19 GF + EdgeConv fwd 2.44 GF; compulsory bytes 2.10 MB + 3.80 MB.
            # 565-640 flop/B: the fp32 roof binds, not HBM (the north star's "HBM roofline" cannot: 189 MB is 24 us at 8 TB/s)
            gf, mb = B_PER_GPU * (1.19 + 2.44), B_PER_GPU * (2.10 + 3.80)
            roof["fwd_knn_edgeconv_eval"] = {"ms": round(fwd_ms, 4), "blocks_per_s": round(B_PER_GPU / (fwd_ms * 1e-3), 1),
                                             "algorithmic_TFLOPps": round(gf / fwd_ms, 2),
                                             "frac_of_fp32_peak": round(gf / fwd_ms / PEAK_FP32_TFLOPS, 4),
                                             "algorithmic_GBps": round(mb / fwd_ms, 1)}
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": "point-cloud blocks/sec (2048 pts, k=20) DGCNN fwd+bwd", "value": round(world * B_PER_GPU / (elapsed / args.steps), 2),
            "unit": "blocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: S3DIS-shaped blocks, batch=32x2048x9 per GPU, k=20, DGCNN+EdgeConv fwd/bwd "
                                   "(train mode, BN batch stats) + grad all-reduce + Adam",
                       "blocks_per_gpu": B_PER_GPU, "points": NPTS, "k": KNN, "parallelism": f"block-sharded dp{world}" + ("+syncbn" if args.sync_bn and world > 1 else "")},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
