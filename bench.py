#!/usr/bin/env python3
"""Throughput of the Effi-MVS+ cost-volume hot path on MI355X (BASELINE.json metric: ref-views/sec).

A "step" is one pass of the hot path (``Effi_MVS_plus.forward_hot``: everything of the reference's
``Effi_MVS_plus.forward`` after the FPN, models/Effi_MVS_plus.py:437-568) over ONE reference view whose
per-stage features and context already sit in HBM.  N > 1: one process per GPU (launched by
``python -m torch.distributed.run``), every rank owns its own shard of reference views (the path shards
by view with no data-path collective); the only collective is the final RCCL gather of the
finished depth / confidence maps to rank 0, which is inside the timed region.

One JSON line on stdout (rank 0).  ``roofline`` is measured live with HIP events around the launches of
the dominant kernel during the timed steps; ``cpu_baseline`` times the CPU oracle (a restatement that
is bitwise-equal to the reference's PyTorch CPU path) on this box's host cores, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (H, W, n_views, ndepths)   -- SURVEY.md section 8(d)
    "cfg1": (128, 160, 4, "8,8,8"),
    "cfg2": (576, 800, 5, "48,8,8"),
    "cfg3": (1184, 1600, 5, "48,8,8"),       # the configuration BASELINE.json's metric is quoted on
    "cfg3b": (1184, 1600, 5, "48,32,8"),
    "cfg4": (1056, 1920, 7, "96,8,8"),
    # the whole DTU evaluation set (22 scans x 49 reference views = 1,078 items, lists/dtu/test.txt) at cfg3's shape, sharded by item
    # over the ranks: its own mode (run_cfg5): images -> pyramids (cached per scan) -> hot path -> batched gather
    "cfg5": (1184, 1600, 5, "48,8,8"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3                 # MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0                # dense bf16 matrix peak; a split-precision product costs three bf16 MFMAs
PEAK_HBM_GBPS = 8000.0
DTYPE = {"split": "f32 (3x3 MFMA convs: products as 3 bf16 partial products hi*hi+hi*lo+lo*hi, f32 accumulate; all else f32)",
         "fp32": "f32",
         "bf16": "bf16 operands in the 3x3 / 3-D MFMA convs (hi*hi only), f32 accumulate; all else f32 -- NOT fp32-grade, own tolerance 1e-2"}


def family_of(key):
    """Kernel TEMPLATE a profile key belongs to: the per-instantiation keys of ops._call (conv2d_k3x3_nt2_epi1, conv2d_k3k1_nt1, ...)
    of one template are one family -- the roofline brackets the family with the largest share of a view, not its largest key."""
    if key.startswith(("conv2d_k3x3", "conv2d_k3k1")):        # conv2d_k3_bf16x3_kernel / _pair_kernel, every NT / MR / epilogue
        return "conv2d_k3_bf16x3"
    if key.startswith("encgen_pair"):                         # conv2d_k3_bf16x3_encgen_pair_kernel: its own template (lookup + 1x1 / 7x7 + 3x3)
        return "encgen_pair"
    return key


def path_work(H, W, N, nd):
    """Algorithmic work of the hot path per reference view, SURVEY.md section 8(d) ("Algorithmic work per reference view"):
    -> (FLOPs, bytes).  cfg3: 233.0 GFLOP, 569 MB; cfg2: 56.7 / 138; cfg3b: 246.7 / 655; cfg4: 269.2 / 795."""
    S, Ds = N - 1, [int(x) for x in nd.split(",")]
    hd, cd, C = (48, 32, 16), (12, 8, 4), (32, 16, 8)
    mac = {48: 714240, 32: 320256, 16: 82176}                 # MACs per pixel of a stage's update block (3 iterations + mask head)
    fl = by = 0.0
    for s in range(3):
        h, w = H // (8 >> s), W // (8 >> s)
        px, D = h * w, Ds[s]
        V = D * px
        fl += S * V * (10 * C[s] + 20)                                                     # warp + correlate + aggregate
        by += 4 * (S * C[s] * px + C[s] * px + D * px + px + S * (H // 8) * (W // 8)) + (8 * S * V if s == 0 else 0)
        if s == 0:                                                                         # 3-D U-Net
            fl += 2 * 4752 * V
            by += 8 * V
        else:                                                                              # the two cross-scale blocks
            fl += 4 * 4104 * D * (h // 2) * (w // 2)
            by += 8 * (2 * V + V / 4)
        fl += 2 * px * mac[hd[s]]                                                          # update block
        by += 4 * px * (2 * hd[s] + cd[s] + 2 * D + 41)
    return fl, by


def run_cfg5(args, torch, dist, rank, world, dev):
    """BASELINE.json cfg5: every (scan, reference view) of an evaluation set, sharded contiguously over the ranks.
    Inside the timed region, per item (effi_mvs_plus_amd.scan_eval.ScanRunner): the synthetic image and feature pyramid of each image
    the rank has not seen in this scan (an image is the source of ~4 other views) on a producer stream; on one of ``--cfg5-slots``
    lanes the reference image written into the slot, one small launch that points the slot's view table at the cached maps, one
    hipGraph replay of context pyramid + hot path, copies of the two output maps into the gather's staging batch; every
    ``--gather-batch`` views one asynchronous gather to rank 0.  Strong scaling: the set is fixed, N ranks split it.  After the timed
    region a sample of the gathered views is recomputed one at a time, eagerly, and must equal the gathered maps bitwise."""
    from common import build_model
    from effi_mvs_plus_amd import _lib, ops, scan_eval, synth
    _lib.lib()
    if args.precision:
        ops.set_precision(args.precision)
    precision = ops.get_precision()
    H, W, N, nd = WORKLOADS["cfg5"]
    if args.cfg5_size:
        W, H = (int(v) for v in args.cfg5_size.lower().split("x"))
    net, _ = build_model(nd, seed=1, device=dev)
    items = scan_eval.build_items(args.cfg5_scans, args.cfg5_images, N - 1)
    _, pm, dv = synth.synth_sample(H, W, N, seed=0)          # one rig for every item: view 0 = the reference, 1.. = its sources
    pm = {k: v.to(dev) for k, v in pm.items()}
    dv = dv.to(dev)
    distributed = world > 1

    def image(scan, img, out=None):                            # a synthetic "decoded image", generated on the device
        g = torch.Generator(device=dev).manual_seed(100003 * scan + img)
        if out is None:
            return torch.rand(1, 3, H, W, device=dev, generator=g)
        return out.uniform_(0.0, 1.0, generator=g)

    with torch.no_grad():
        runner = scan_eval.ScanRunner(net, image, (10 ** 6, 0, tuple(range(1, N))), pm, dv, slots=args.cfg5_slots)
        torch.cuda.synchronize()
        # untimed warm-up on a scan of its own (same shapes; its cache entries are dropped by the first timed item)
        warm = [(10 ** 6, it[1], it[2]) for it in scan_eval.build_items(1, min(args.cfg5_images, max(args.warmup, N + 1)), N - 1)]
        for j, it in enumerate(warm[:max(1, args.warmup)]):
            with torch.cuda.stream(runner.lanes[j % runner.slots]):
                runner(it, j % runner.slots)
        torch.cuda.synchronize()
        runner.cache.hits = runner.cache.misses = runner.cache.max_entries = 0
        runner.n_images_prepared = 0
        torch.cuda.reset_peak_memory_stats()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res, n_mine = scan_eval.run_scans(items, runner, gather_batch=args.gather_batch, dst=0, to_host=(distributed and args.backend != "nccl"))
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        feats = runner.cache
        # a sample of this rank's views again, one at a time, eagerly, from freshly computed pyramids: bitwise the gathered maps?
        n_checked = n_bad = 0
        if rank == 0 and res is not None and args.cfg5_check > 0:
            lo, hi = scan_eval.shard_bounds(len(items), 0, world)
            picks = sorted({lo + (hi - 1 - lo) * k // max(args.cfg5_check - 1, 1) for k in range(args.cfg5_check)})
            br0 = ops.get_branches()
            ops.set_branches(False)
            for i_ in picks:
                scan, ref, srcs = items[i_]
                f = [net.feature(image(scan, v)) for v in (ref,) + tuple(srcs)]
                o = net.forward_hot(f, net.cnet_depth(image(scan, ref)), pm, dv)
                ok_d = torch.equal(o["depth"][-1][0], res["depth"][i_].to(dev))
                ok_c = torch.equal(o["photometric_confidence"][0], res["confidence"][i_].to(dev))
                n_checked += 1
                n_bad += 0 if (ok_d and ok_c) else 1
            ops.set_branches(br0)
    peak = torch.cuda.max_memory_allocated()
    stats = torch.tensor([dt, float(peak), float(feats.misses), float(feats.hits), float(n_mine)], dtype=torch.float64,
                         device=dev if (not distributed or args.backend == "nccl") else "cpu")
    if distributed:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    else:
        mx = sm = stats
    if rank == 0:
        dt = float(mx[0])
        ok = res is not None and res["depth"].shape[0] == len(items) and bool(torch.isfinite(res["depth"]).all()) \
            and bool(torch.isfinite(res["confidence"]).all())
        print(json.dumps({
            "metric": "ref-views/sec (whole evaluation set: pyramids + cost-volume hot path + batched gather to rank 0)",
            "value": len(items) / dt, "unit": "views/s", "n_gpus": world, "steps": len(items), "warmup": max(1, args.warmup),
            "ms_per_step": dt / len(items) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": DTYPE[precision], "data": "synthetic",
            "config": {"workload": f"cfg5: {args.cfg5_scans} scans x {args.cfg5_images} reference views = {len(items)} items at {W}x{H}, N={N} "
                                   f"(S={N - 1} sources from a ring pair list), ndepths={nd}, GRU iters 3,3,3, seeded-random weights, synthetic images "
                                   "generated on the device inside the timed region",
                       "launch": f"{runner.slots} reference views in flight (one lane = stream + captured hipGraph of context pyramid + hot path "
                                 "each); feature pyramids of uncached images eagerly on a producer stream; the graph reads the item's cached "
                                 "feature maps through a device pointer table (no copy into static inputs)",
                       "parallelism": (f"items sharded contiguously over {world} ranks, {'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal)'} "
                                       f"gather of depth + confidence to rank 0 every {args.gather_batch} views, asynchronous, inside the timed region")
                                      if world > 1 else "single GPU (same code path: staging batches, no collective)"},
            "seconds_for_the_set": dt, "views_per_rank_max": int(mx[4]),
            "feature_cache": {"pyramids_computed_all_ranks": int(sm[2]), "pyramids_reused_all_ranks": int(sm[3]),
                              "note": "an image's feature pyramid is computed once per rank and scan and reused while it is a source view"},
            "hbm_high_water_bytes_max_over_ranks": int(mx[1]),
            "gathered_on_rank0": {"depth": list(res["depth"].shape), "confidence": list(res["confidence"].shape), "finite": ok},
            "recomputed_one_at_a_time": {"views": n_checked, "differing_bitwise": n_bad,
                                         "note": "rank 0's sample of gathered views against an eager single-stream forward from fresh pyramids"},
            "scaling_curve": "NOT measured by this run: one run is one N.  Comparing N = 1, 2, 4, 8 needs the driver's multi-GPU node; "
                             "no efficiency is claimed here" if world == 1 else "one point of the curve (this N); efficiency is the driver's to compute",
        }))
        if not ok:
            sys.exit(4)
        if n_bad:
            print(f"[bench] {n_bad} of {n_checked} recomputed views differ bitwise from the gathered maps", file=sys.stderr)
            sys.exit(3)


def self_launch(n, backend, ndev):
    """One process per GPU on this node through torch.distributed.run (the same command line the driver uses); returns the
    launcher's exit code.  Refuses (non-zero) when the RCCL path would need more devices than the node has."""
    import socket
    import subprocess
    if backend == "nccl" and ndev < n:
        print(f"[bench] --gpus {n} needs {n} GPUs for the RCCL path, this node has {ndev}; use --backend gloo to rehearse the "
              "multi-rank plumbing on fewer devices", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout of this process carries ONE JSON line: whatever else the ranks' libraries write there (gloo announces its
    # connections on stdout) goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-views", type=int, default=2)
    ap.add_argument("--torch-baseline-views", type=int, default=3,
                    help="also time the reference-style composite PyTorch-ROCm path (the oracle's op sequence run on the GPU); 0 = skip")
    ap.add_argument("--profile-key", default=None, help="kernel key to bracket with events (default: auto = largest total time)")
    ap.add_argument("--no-whole-forward", action="store_true", help="skip the secondary whole-forward timings (profiling runs)")
    ap.add_argument("--precision", default=None, choices=["split", "fp32", "bf16"],
                    help="arithmetic of the 3x3 MFMA convolutions (default: the library default, EFFI_MVS_PRECISION or 'split')")
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"],
                    help="graph (default): the step copies its inputs into the static buffers of a captured hipGraph of the hot path "
                         "and replays it (one launch); eager: ~100 launches enqueued from Python (about as long on the host as on the GPU)")
    ap.add_argument("--pipelined", type=int, default=3,
                    help="secondary measurement: throughput with this many independent views in flight (0/1 = skip)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="graph launch only: reference views in flight at once, each on its own stream and input slot, its graph "
                         "captured with the pass's two internal streams (throughput metric: while one view is in its low-resolution "
                         "stages, which cannot fill 256 CUs, the others use them); 1 = strictly one view after the other on a linear "
                         "graph.  With more than one the single-stream figure is measured right after and reported as 'single_stream'")
    ap.add_argument("--graph-branches", default="auto", choices=["auto", "0", "1"],
                    help="views in flight: capture each view's graph with the pass's side stream (1) or as a linear one-stream graph (0); "
                         "auto = the measured better form (see DESIGN.md section 6)")
    ap.add_argument("--no-other-precision", action="store_true", help="skip the secondary run in the other conv arithmetic (profiling runs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1: nccl (= RCCL over xGMI, the real path) or gloo (rehearsal of the "
                         "multi-rank plumbing on a box with fewer GPUs than ranks; ranks then share devices)")
    ap.add_argument("--cfg5-scans", type=int, default=22, help="--workload cfg5: scans in the evaluation set (DTU test list: 22)")
    ap.add_argument("--cfg5-images", type=int, default=49, help="--workload cfg5: images = reference views per scan (DTU: 49)")
    ap.add_argument("--gather-batch", type=int, default=8, help="--workload cfg5: views per asynchronous gather to rank 0")
    ap.add_argument("--cfg5-slots", type=int, default=4, help="--workload cfg5: reference views in flight per rank (lanes); 3 / 4 / 5 on one box: "
                                                               "477 / 482 / 465 views/s (profiles/r04_ag_cfg5_slots.txt)")
    ap.add_argument("--cfg5-check", type=int, default=6, help="--workload cfg5: gathered views recomputed one at a time after the timed "
                                                               "region and compared bitwise (0 = skip)")
    ap.add_argument("--cfg5-size", default=None, help="--workload cfg5: WxH instead of 1600x1184 (rehearsals)")
    ap.add_argument("--fusion-torch-baseline", action="store_true",
                    help="also time the reference's fusion op sequence on the GPU (element-wise 3x3 algebra instead of the 19 M-batch "
                         "GEMM that faulted in round 1); opt-in, never part of the default run")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # ``python bench.py --gpus N`` without an external launcher: start the N ranks ourselves, BEFORE anything in this
        # process touches the GPU (device_count() does not initialise it), and pass their exit code on
        sys.exit(self_launch(args.gpus, args.backend, torch.cuda.device_count()))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1
    ndev = torch.cuda.device_count()
    if distributed and args.backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks need {world} GPUs for the RCCL path (found {ndev}); use --backend gloo to rehearse")
    dev_index = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)     # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend="gloo")

    if args.workload == "cfg5":
        run_cfg5(args, torch, dist, rank, world, dev)
        if distributed:
            dist.destroy_process_group()
        return

    from common import build_model
    from effi_mvs_plus_amd import _lib, ops, shard, synth

    _lib.lib()                                   # fail loudly if the HIP library is missing
    if args.precision:
        ops.set_precision(args.precision)
    precision = ops.get_precision()
    H, W, N, nd = WORKLOADS[args.workload]
    net, sd = build_model(nd, seed=1, device=dev)

    # ---- inputs: synthetic rig, features of the stock FPN, everything resident in HBM --------------
    n_scenes = 2                                  # alternate two synthetic "views" so steps are not identical
    inputs = []
    with torch.no_grad():
        for i in range(n_scenes):
            imgs, pm, dv = synth.synth_sample(H, W, N, seed=1000 * rank + i)
            imgs = imgs.to(dev)
            feats = [net.feature(imgs[:, v]) for v in range(N)]
            ctx = net.cnet_depth(imgs[:, 0])
            inputs.append((feats, ctx, {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
            del imgs
    torch.cuda.synchronize()

    graphed = None
    graph_fallback = None
    graph_branches = False
    if args.launch == "graph":
        from effi_mvs_plus_amd.graph import HotPathGraph
        # double-buffered input slots (as a producer of features would fill them); the two synthetic views are loaded
        # into the slots before the timed region -- "inputs resident in HBM" -- and a step replays the slot's graph
        n_slots = max(n_scenes, args.in_flight)
        try:
            if args.in_flight > 1 and args.graph_branches != "0":    # each view's graph keeps the pass's side stream (auto: measured better)
                ops.set_branches(True)
            graphed = HotPathGraph(net, *inputs[0], slots=n_slots)
            for i in range(n_slots):
                graphed.load(i, *inputs[i % n_scenes])
            torch.cuda.synchronize()
        except Exception as exc:      # capture refused by the runtime (never seen on one GPU): keep measuring, eagerly
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches",
                  file=sys.stderr)
            graphed = None
            graph_fallback = f"{type(exc).__name__}: {exc}"
            torch.cuda.synchronize()
        finally:
            graph_branches = ops.get_branches()
            ops.set_branches(False)               # everything eager below (discovery, per-kernel events, stage marks) is single-stream
        lanes = [torch.cuda.Stream() for _ in range(max(1, args.in_flight))]

    def step(i):
        f, c, p, d = inputs[i % n_scenes]
        if graphed is not None and ops.get_profile() is None and ops.get_precision() == precision:
            return graphed.replay(i % n_slots)       # one graph launch
        return net.forward_hot(f, c, p, d)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # ---- discovery pass (untimed): which kernel dominates? ------------------------------------
        for i in range(max(1, args.warmup - 1)):
            step(i)
        prof = ops.KernelProfile()
        br0 = ops.get_branches()
        ops.set_branches(False)                   # one stream: per-kernel durations are not inflated by overlap
        ops.set_profile(prof)
        step(0)
        ops.set_profile(None)
        ops.set_branches(br0)
        disc = prof.summary()
        fam_ms = {}
        for k_, v_ in disc.items():
            fam_ms[family_of(k_)] = fam_ms.get(family_of(k_), 0.0) + v_["ms"]
        key = args.profile_key or max(fam_ms, key=lambda k: fam_ms[k])      # the kernel TEMPLATE with the largest share of a view
        fam_keys = sorted(k_ for k_ in disc if family_of(k_) == key or k_ == key)
        step(1)                                   # last warm-up step, no instrumentation
        if graphed is not None and args.in_flight > 1:
            # ... and W warm-up steps launched exactly as the timed ones are: slot i's graph on lane i.  A stream's hardware queue
            # is created at its first use and a graph's first launch into a stream uploads it there; with the warm-up on the
            # current stream only, both fell into the timed region (20 steps: 590 views/s; the same steps after this warm-up:
            # see DESIGN.md section 6)
            cur = torch.cuda.current_stream()
            for st_ in lanes:
                st_.wait_stream(cur)
            for i in range(max(args.warmup, len(lanes))):
                with torch.cuda.stream(lanes[i % len(lanes)]):
                    step(i)
            for st_ in lanes:
                cur.wait_stream(st_)
            torch.cuda.synchronize()

        # ---- timed region: exactly K steps + the final gather ---------------------------------------
        # eager launches: the dominant kernel's launches are bracketed with HIP events inside the timed region;
        # graph replay has no per-kernel events, there the same K steps are repeated eagerly right after it
        prof = ops.KernelProfile(keys=fam_keys)
        if graphed is None:
            ops.set_profile(prof)
        finals, confs = [], []
        barrier()
        t0 = time.perf_counter()
        if graphed is not None and args.in_flight > 1:
            # views i and i+1 are independent: step i runs on stream i % in_flight (its own input slot and graph)
            cur = torch.cuda.current_stream()
            for st_ in lanes:
                st_.wait_stream(cur)
            for i in range(args.steps):
                with torch.cuda.stream(lanes[i % len(lanes)]):
                    out = step(i)
                    finals.append(out["depth"][-1].clone())          # static output buffers: keep copies
                    confs.append(out["photometric_confidence"].clone())
            for st_ in lanes:
                cur.wait_stream(st_)
        else:
            for i in range(args.steps):
                out = step(i)
                if graphed is not None:               # static output buffers: keep copies (inside the timed region)
                    finals.append(out["depth"][-1].clone())
                    confs.append(out["photometric_confidence"].clone())
                else:
                    finals.append(out["depth"][-1])
                    confs.append(out["photometric_confidence"])
        if distributed:
            # the path's only collective: ONE RCCL gather per tensor of this rank's finished maps to rank 0
            to = (lambda t_: t_) if args.backend == "nccl" else (lambda t_: t_.cpu())   # gloo gathers host tensors
            shard.gather_maps(to(torch.cat(finals)), args.steps * world, dst=0)
            shard.gather_maps(to(torch.cat(confs)), args.steps * world, dst=0)
        barrier()
        dt = time.perf_counter() - t0
        ops.set_profile(None)
        if graphed is not None:
            br = ops.get_branches()
            ops.set_branches(False)                   # one stream: the bracketed launch runs alone, its duration is its own
            ops.set_profile(prof)
            for i in range(args.steps):
                step(i)                               # eager (step() never replays while a profile is installed)
            ops.set_profile(None)
            ops.set_branches(br)
            torch.cuda.synchronize()
        # the same K steps strictly one view after the other (a view's latency): linear graphs, one stream, outputs cloned per step
        single_stream = None
        if graphed is not None and args.in_flight > 1 and rank == 0:
            from effi_mvs_plus_amd.graph import HotPathGraph
            try:
                sg = HotPathGraph(net, *inputs[0], slots=n_scenes)
                for i in range(n_scenes):
                    sg.load(i, *inputs[i % n_scenes])
                for i in range(n_scenes):
                    sg.replay(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(args.steps):
                    o = sg.replay(i % n_scenes)
                    keep = (o["depth"][-1].clone(), o["photometric_confidence"].clone())
                torch.cuda.synchronize()
                ds = time.perf_counter() - t1
                same = all(bool(torch.equal(a_, b_)) for a_, b_ in zip(sg.replay(0)["depth"], graphed.replay(0)["depth"]))
                # every view the TIMED region produced (cloned while the other views were in flight) against the single-stream result of its scene
                ref_final = [sg.replay(sc)["depth"][-1].clone() for sc in range(n_scenes)]
                ref_conf = [sg.replay(sc)["photometric_confidence"].clone() for sc in range(n_scenes)]
                torch.cuda.synchronize()
                n_bad = sum(1 for i, (f_, c_) in enumerate(zip(finals, confs))
                            if not (torch.equal(f_, ref_final[(i % n_slots) % n_scenes]) and torch.equal(c_, ref_conf[(i % n_slots) % n_scenes])))
                single_stream = {"value": args.steps / ds, "unit": "views/s", "ms_per_view": ds / args.steps * 1e3, "steps": args.steps,
                                 "bitwise_equal_to_the_in_flight_graphs": same,
                                 "timed_in_flight_views_differing_from_single_stream": n_bad,
                                 "note": "one view after the other: linear hipGraph replay on one stream (a view's latency; rank 0)"}
                del sg, keep
            except Exception as exc:
                single_stream = {"error": f"{type(exc).__name__}: {exc}"}
    # ms per cost-volume stage (the second half of BASELINE.json's metric): HIP events at the stage boundaries of eager
    # single-stream passes (stage k = its cost volume + regularisation / cross-scale blocks + its three GRU iterations + upsampling;
    # stage 1 also carries the preparation of all stages)
    stage_ms = stage_marks_ms = None
    if rank == 0:
        with torch.no_grad():
            br = ops.get_branches()
            ops.set_branches(False)
            acc_ms, n_pass = {}, 5
            net.forward_hot(*inputs[0])               # untimed: the allocator's blocks of an eager pass are warm again
            torch.cuda.synchronize()
            for i in range(n_pass):
                marks = []
                ops.set_marks(marks)
                net.forward_hot(*inputs[i % n_scenes])
                ops.set_marks(None)
                torch.cuda.synchronize()
                for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                    acc_ms[n1] = acc_ms.get(n1, 0.0) + e0.elapsed_time(e1)
            stage_marks_ms = {k: v / n_pass for k, v in acc_ms.items()}
            # the same spans as the sum of their kernels' own durations (per-launch events): what a stage costs when its kernels run
            # back to back, as in the replay -- the event marks above also contain whatever the host failed to enqueue in time
            busy = {}
            for i in range(n_pass):
                pr = ops.KernelProfile()
                ops.set_profile(pr)
                net.forward_hot(*inputs[i % n_scenes])
                ops.set_profile(None)
                for k, v in pr.busy_ms_per_mark().items():
                    busy[k] = busy.get(k, 0.0) + v
            ops.set_branches(br)
            stage_ms = {k: v / n_pass for k, v in busy.items() if k != "begin"}
    if distributed:
        tmax = torch.tensor([dt], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kinst = prof.summary()                        # per instantiation (profile key) of the bracketed kernel template
    ksum = {f_: sum(v_[f_] for v_ in kinst.values()) for f_ in ("launches", "ms", "flops", "bytes")}

    result = None
    bf16_outputs = None
    if rank == 0:
        views = args.steps * world
        avg_ms = ksum["ms"] / ksum["launches"]
        flops_per_launch = ksum["flops"] / ksum["launches"]
        bytes_per_launch = ksum["bytes"] / ksum["launches"]
        intensity = flops_per_launch / max(bytes_per_launch, 1.0)
        # split-precision kernels: three bf16 MFMAs per fp32-equivalent product (plain bf16 operands: one)
        split_keys = ("conv2d_k3", "conv3d_x3", "conv3d_roll", "deconv3d_x3")
        per_product = {"split": 3.0, "bf16": 1.0}.get(precision)
        mfma_peak = PEAK_BF16_MFMA_TFLOPS / per_product if (per_product and key.startswith(split_keys)) else PEAK_FP32_MFMA_TFLOPS
        tflops = flops_per_launch / (avg_ms * 1e-3) / 1e12
        gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        if key.startswith(("conv", "deconv")) and intensity > mfma_peak * 1e12 / (PEAK_HBM_GBPS * 1e9):
            roof = {"bound": "mfma", "achieved": tflops, "peak": mfma_peak, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        # both roofs of the bracketed kernel, whichever binds by intensity: FLOP-based against the matrix peak of its arithmetic, byte-based
        # against HBM (algorithmic FLOPs / bytes of all its launches in a view / their summed duration)
        roof["frac_mfma"] = tflops / mfma_peak
        roof["frac_hbm"] = gbps / PEAK_HBM_GBPS
        roof["achieved_tflops"], roof["achieved_gbps"], roof["mfma_peak_tflops"] = tflops, gbps, mfma_peak
        roof["intensity_flop_per_byte"] = intensity
        roof["instances"] = {k_: {"launches": v_["launches"], "avg_launch_us": v_["ms"] / v_["launches"] * 1e3,
                                  "tflops": v_["flops"] / (v_["ms"] * 1e-3) / 1e12, "gbps": v_["bytes"] / (v_["ms"] * 1e-3) / 1e9}
                             for k_, v_ in sorted(kinst.items(), key=lambda kv: -kv[1]["ms"]) if v_["launches"]}
        # HBM bytes from the PMC counters (FETCH_SIZE / WRITE_SIZE), collected offline in separate rocprofv3 --pmc passes of this
        # same command and summarised by tools/pmc_traffic.py (profiles/pmc_traffic_<workload>.json): per launch of the bracketed
        # kernel (launch-weighted over its instantiations), and for the whole view against the path's algorithmic bytes
        roof["traffic"] = None
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}.json")
        pmc = None
        if os.path.exists(tpath):
            with open(tpath) as f:
                pmc = json.load(f)
            tb = tl_ = 0.0
            for k_, v_ in kinst.items():
                tr = pmc.get(k_)
                if tr and v_["launches"]:
                    tb += (tr["fetch_bytes"] + tr["write_bytes"]) * v_["launches"]
                    tl_ += v_["launches"]
            if tl_:
                roof["traffic"] = tb / tl_
                roof["traffic_detail"] = {k_: pmc[k_] for k_ in kinst if k_ in pmc}
        roof["kernel"] = key
        roof["events"] = ("HIP events around the kernel's launches inside the timed region" if graphed is None else
                          "HIP events around the kernel's launches in the same K steps enqueued eagerly on one stream right after the "
                          "timed graph replays (a replay has no per-kernel events; same kernels, same launch configuration)")
        roof["avg_launch_ms"] = avg_ms
        # what an event pair with NOTHING between its records reads on this stream: the bracket's own cost, contained in avg_launch_ms
        # (rocprofv3's kernel-trace average of the same launches is lower by about this much; 'achieved' keeps the conservative raw figure)
        gaps = []
        for _ in range(50):
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            g1.record()
            gaps.append((g0, g1))
        torch.cuda.synchronize()
        roof["empty_event_bracket_ms"] = sorted(a_.elapsed_time(b_) for a_, b_ in gaps)[len(gaps) // 2]
        net_ms = max(avg_ms - roof["empty_event_bracket_ms"], 1e-6)
        roof["frac_without_the_empty_bracket"] = roof["frac"] * avg_ms / net_ms      # what rocprofv3's per-kernel average implies
        roof["launches_timed"] = ksum["launches"]
        roof["algorithmic_flops_per_launch"] = flops_per_launch
        roof["algorithmic_bytes_per_launch"] = bytes_per_launch
        total_ms = sum(v["ms"] for v in disc.values())
        roof["share_of_kernel_time"] = fam_ms.get(key, disc.get(key, {"ms": 0.0})["ms"] if key in disc else 0.0) / max(total_ms, 1e-9)
        # the whole path of one view: algorithmic FLOPs and bytes over the single-stream time of a view
        p_fl, p_by = path_work(H, W, N, nd)
        ss_ms = (single_stream or {}).get("ms_per_view") or dt / args.steps * 1e3
        path_peak = PEAK_BF16_MFMA_TFLOPS / per_product if per_product else PEAK_FP32_MFMA_TFLOPS
        roof["path"] = {"flops": p_fl, "bytes": p_by, "ms_single_stream": ss_ms,
                        "tflops": p_fl / (ss_ms * 1e-3) / 1e12, "gbps": p_by / (ss_ms * 1e-3) / 1e9,
                        "frac_mfma": p_fl / (ss_ms * 1e-3) / 1e12 / path_peak, "frac_hbm": p_by / (ss_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                        "mfma_peak_tflops": path_peak,
                        "note": "algorithmic work of one reference view (SURVEY.md 8(d)) / time of one view, one view after the other"}
        roof["traffic_per_view"] = None
        if pmc and pmc.get("_meta", {}).get("views"):
            tv = sum((v_["fetch_bytes"] + v_["write_bytes"]) * v_["launches_profiled"] for k_, v_ in pmc.items() if k_ != "_meta")
            tv /= pmc["_meta"]["views"]
            roof["traffic_per_view"] = {"bytes": tv, "ratio_to_algorithmic": tv / p_by,
                                        "note": "sum over ALL kernels of a view of (FETCH_SIZE x calibration + WRITE_SIZE) x launches / views "
                                                "(tools/pmc_traffic.sh); the excess over the path's algorithmic bytes is what the layer-by-layer "
                                                "update block writes and reads back"}
        result = {
            "metric": "ref-views/sec (cost-volume hot path: warp + cost volume + 3-D regularisation + cascaded GRU refinement)",
            "value": views / dt, "unit": "views/s", "n_gpus": (dist.get_world_size() if distributed else 1),       # as the process group reports it
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[precision], "data": "synthetic",
            "config": {"workload": f"{args.workload}: DTU-shaped {W}x{H}, N={N} views (S={N - 1} sources), 3-stage cascade "
                                   f"ndepths={nd}, GRU iters 3,3,3, seeded-random weights, features of the stock FPN resident in HBM",
                       "launch": ("hipGraph replay of the captured hot path (" + ("two streams" if graph_branches else "one stream, linear graph")
                                  + ") on double-buffered static input slots that hold the "
                                  f"synthetic views; each step replays one slot and clones the outputs it keeps, inside the timed region; "
                                  f"{max(1, args.in_flight)} independent view(s) in flight, one stream each"
                                  if graphed is not None else "eager: every kernel enqueued from Python"),
                       "parallelism": f"view-sharded x{world}, {'RCCL' if args.backend == 'nccl' else 'gloo (rehearsal)'} gather of "
                                      f"depth+confidence to rank 0 inside the timed region" if world > 1 else "single GPU",
                       # what 'value' is and is not: views in flight, split-precision products; the other two figures beside it
                       "views_in_flight": (max(1, args.in_flight) if graphed is not None else 1),
                       "single_stream_ms": (single_stream or {}).get("ms_per_view"),
                       "single_stream_views_per_s": (single_stream or {}).get("value"),
                       "fp32_strict_views_per_s": None, "fp32_strict_ms_per_view": None,
                       "arithmetic": precision},
            "in_flight": (max(1, args.in_flight) if graphed is not None else 1),
            # one view after the other (a view's latency) next to the headline's views in flight; strict-fp32 figure filled in below
            "value_single_stream": (single_stream or {}).get("value"),
            "ms_per_view_single_stream": (single_stream or {}).get("ms_per_view"),
            "value_fp32": None,
            **({"single_stream": single_stream} if single_stream is not None else {}),
            "ms_per_cost_volume_stage": stage_ms,
            "ms_per_cost_volume_stage_note": "sum of the stage's kernel durations (per-launch HIP events, eager single-stream passes) = its "
                                             "cost when the kernels run back to back as in the graph replay; "
                                             "'ms_per_cost_volume_stage_event_marks' = events at the stage boundaries of eager passes, which "
                                             "also contain the gaps of a host-bound eager launch sequence",
            "ms_per_cost_volume_stage_event_marks": stage_marks_ms,
            **({"graph_fallback": graph_fallback} if graph_fallback else {}),
            "roofline": roof,
            "kernel_breakdown_ms": {k: round(v["ms"], 4) for k, v in sorted(disc.items(), key=lambda kv: -kv[1]["ms"])},
        }

    # ---- secondary (rank 0, N = 1, outside the timed region): the same K steps with the other conv arithmetic -- measured the
    # same way as the headline (its own captured graph, K replays + output clones) and, for reference, eagerly -- and the
    # distance between the two modes' final depth maps (normalised by the depth range, as the parity tests do)
    if rank == 0 and world == 1 and not args.no_other_precision:
        other = "fp32" if precision != "fp32" else "split"
        with torch.no_grad():
            ref_out = step(0)["depth"][-1].clone()
            ops.set_precision(other)
            alt_out = step(0)["depth"][-1].clone()              # eager: step() replays only in the headline precision
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            torch.cuda.synchronize()
            dt_other = time.perf_counter() - t0
            other_graph = None
            if graphed is not None:
                from effi_mvs_plus_amd.graph import HotPathGraph
                try:
                    og = HotPathGraph(net, *inputs[0], slots=n_scenes)          # captured while ``other`` is the precision
                    for i in range(n_scenes):
                        og.load(i, *inputs[i % n_scenes])
                    for i in range(2):
                        og.replay(i % n_scenes)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    keep = []
                    for i in range(args.steps):
                        o = og.replay(i % n_scenes)
                        keep.append((o["depth"][-1].clone(), o["photometric_confidence"].clone()))
                    torch.cuda.synchronize()
                    dt_og = time.perf_counter() - t0
                    same = bool(torch.equal(og.replay(0)["depth"][-1], alt_out))
                    other_graph = {"value": args.steps / dt_og, "unit": "views/s", "ms_per_step": dt_og / args.steps * 1e3,
                                   "in_flight": 1, "replay_bitwise_equal_to_eager": same}
                    del og, keep
                except Exception as exc:
                    other_graph = {"error": f"{type(exc).__name__}: {exc}"}
            ops.set_precision(precision)
        rng = synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM
        diff = (ref_out - alt_out).abs() / rng
        if graphed is not None:                      # the default arithmetic without the graph, for reference
            with torch.no_grad():
                ops.set_profile(ops.KernelProfile(keys=[]))          # an installed profile keeps step() on the eager path
                step(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    step(i)
                torch.cuda.synchronize()
                dt_eager = time.perf_counter() - t0
                ops.set_profile(None)
            result["eager_launch"] = {"value": args.steps / dt_eager, "unit": "views/s", "ms_per_step": dt_eager / args.steps * 1e3,
                                      "note": "same kernels enqueued from Python (~100 launches per view): the host needs about as long as the GPU"}
        # BASELINE.json's literal "bf16 (MFMA 3D-conv path)" configuration: the same kernels compiled with plain bf16 operands (hi*hi
        # only, fp32 accumulation), timed like the headline.  Its own line with its own tolerance; never the headline.
        if precision != "bf16" and graphed is not None:
            from effi_mvs_plus_amd.graph import HotPathGraph
            try:
                ops.set_precision("bf16")
                with torch.no_grad():
                    bg = HotPathGraph(net, *inputs[0], slots=n_scenes)
                    for i in range(n_scenes):
                        bg.load(i, *inputs[i % n_scenes])
                    bf16_outputs = [d_.clone() for d_ in bg.replay(0)["depth"]]
                    bg.replay(1)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    keep = []
                    for i in range(args.steps):
                        o = bg.replay(i % n_scenes)
                        keep.append((o["depth"][-1].clone(), o["photometric_confidence"].clone()))
                    torch.cuda.synchronize()
                    dtb = time.perf_counter() - t0
                result["bf16_operands"] = {"value": args.steps / dtb, "unit": "views/s", "ms_per_step": dtb / args.steps * 1e3, "dtype": DTYPE["bf16"],
                                           "launch": "hipGraph replay, as the headline"}
                del bg, keep
            except Exception as exc:
                result["bf16_operands"] = {"error": f"{type(exc).__name__}: {exc}"}
                bf16_outputs = None
            finally:
                ops.set_precision(precision)
            torch.cuda.empty_cache()
        if other == "fp32" and other_graph and "value" in other_graph:
            result["value_fp32"] = other_graph["value"]           # exact fp32 products everywhere, one view after the other (graph replay)
            # ... and inside ``config`` (the driver keeps config and drops extra keys): nobody should read the headline alone
            result["config"]["fp32_strict_views_per_s"] = other_graph["value"]
            result["config"]["fp32_strict_ms_per_view"] = other_graph["ms_per_step"]
        result["other_precision"] = {"mode": other, "dtype": DTYPE[other],
                                     "graph_replay": other_graph,        # one view after the other: like-for-like with 'single_stream' (or the headline when --in-flight 1)
                                     "eager": {"value": args.steps / dt_other, "unit": "views/s", "ms_per_step": dt_other / args.steps * 1e3},
                                     "final_depth_diff_between_modes": {"mean_norm": float(diff.mean()),
                                                                        "p99_norm": float(diff.flatten().kthvalue(int(0.99 * diff.numel())).values),
                                                                        "max_norm": float(diff.max())}}
        torch.cuda.empty_cache()

    # ---- secondary (rank 0, N = 1): throughput with several independent views in flight (their graphs on separate streams, the
    # second stream inside each graph enabled): bubbles of one view are filled by kernels of the others.  Not the headline: a
    # view's latency rises to ~in_flight x ms_per_view.
    if rank == 0 and world == 1 and graphed is not None and args.pipelined > 1 and args.in_flight <= 1 and not args.no_whole_forward:
        from effi_mvs_plus_amd.graph import HotPathGraph
        br0 = ops.get_branches()
        ops.set_branches(True)
        try:
            pg = HotPathGraph(net, *inputs[0], slots=args.pipelined)
            for i in range(args.pipelined):
                pg.load(i, *inputs[i % n_scenes])
            plan = [torch.cuda.Stream() for _ in range(args.pipelined)]
            nst = max(args.steps, 3 * args.pipelined)

            def run_pipelined(n):
                cur = torch.cuda.current_stream()
                for st_ in plan:
                    st_.wait_stream(cur)
                keep = []
                for i in range(n):
                    with torch.cuda.stream(plan[i % len(plan)]):
                        o = pg.replay(i % args.pipelined)
                        keep.append((o["depth"][-1].clone(), o["photometric_confidence"].clone()))
                for st_ in plan:
                    cur.wait_stream(st_)
                return keep

            with torch.no_grad():
                run_pipelined(args.pipelined)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_pipelined(nst)
                torch.cuda.synchronize()
                tp = (time.perf_counter() - t0) / nst
            result["pipelined"] = {"views_in_flight": args.pipelined, "value": 1.0 / tp, "unit": "views/s", "ms_per_view": tp * 1e3,
                                   "steps": nst, "note": "one captured graph (two streams) per input slot, replayed on "
                                                         f"{args.pipelined} streams round-robin; outputs cloned per step as in the headline run"}
            del pg
        except Exception as exc:
            result["pipelined"] = {"error": f"{type(exc).__name__}: {exc}"}
        ops.set_branches(br0)
        torch.cuda.empty_cache()

    # ---- secondary (rank 0, N = 1, outside the timed region): the whole forward as the reference's drivers time it
    # (test_dtu_dypcd.py:437-442: images -> 13 depth maps), with the feature pyramid on the HIP kernels (scope row n1)
    # and, for comparison, with the stock PyTorch-ROCm pyramid in front of the same hot path
    if rank == 0 and world == 1 and not args.no_whole_forward:
        imgs, pm_c, dv_c = synth.synth_sample(H, W, N, seed=0)
        imgs = imgs.to(dev)
        pm_d = {k: v.to(dev) for k, v in pm_c.items()}
        dv_d = dv_c.to(dev)

        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        def fwd_torch_fpn():
            f = [net.feature.forward_torch(imgs[:, v]) for v in range(N)]
            return net.forward_hot(f, net.cnet_depth.forward_torch(imgs[:, 0]), pm_d, dv_d)

        with torch.no_grad():
            fg_ms = None
            if graphed is not None:
                from effi_mvs_plus_amd.graph import ForwardGraph
                fg = ForwardGraph(net, imgs, pm_d, dv_d)
                fg_ms = timed(fg.replay)
                del fg
            full_ms = timed(lambda: net(imgs, pm_d, dv_d))
            fpn_ms = timed(lambda: ([net.feature(imgs[:, v]) for v in range(N)], net.cnet_depth(imgs[:, 0])))
            full_torch_fpn_ms = timed(fwd_torch_fpn)
        best_ms = full_ms if fg_ms is None else min(fg_ms, full_ms)
        result["whole_forward"] = {"ms_per_view": best_ms, "views_per_s": 1e3 / best_ms, "ms_per_view_graph_replay": fg_ms,
                                   "ms_per_view_eager": full_ms, "feature_pyramids_ms": fpn_ms,
                                   "ms_per_view_with_stock_pytorch_pyramid": full_torch_fpn_ms,
                                   "note": "images resident in HBM -> 13 depth maps + confidence; N feature nets + 1 context net + hot path"}
        del imgs

    # ---- secondary (rank 0, N = 1): scope row n3, the consumer of the path's depth maps -- dynamic geometric-consistency
    # filter + depth averaging (misc/fusion.py, test_tank.py:466-512) on full-resolution synthetic depth maps
    if rank == 0 and world == 1 and not args.no_whole_forward:
        from oracle import effi_oracle as Of
        Vf = 10
        dmaps, fcams = synth.synth_depth_maps(H, W, Vf + 1, seed=4)
        dmaps, fcams = dmaps.to(dev), fcams.to(dev)
        fconf = torch.rand(H, W, device=dev)
        fargs = (dmaps[0].contiguous(), dmaps[1:].contiguous(), fcams[0].contiguous(), fcams[1:].contiguous(), fconf, 0.3, 2, 4.0, 1.3)
        with torch.no_grad():
            hip_ms = timed(lambda: ops.fusion_dynamic_filter(*fargs), n=10)
        fbytes = 4.0 * H * W * (1 + Vf + 1 + 1 + 3) + 3.0 * H * W
        result["fusion_filter"] = {"views_per_s": 1e3 / hip_ms, "ms_per_ref_view": hip_ms, "src_views": Vf,
                                   "algorithmic_GBps": fbytes / hip_ms / 1e6,
                                   "note": f"{W}x{H} depth maps, {Vf} source views, dh_view_num 2, one fused kernel per reference view"}
        if args.fusion_torch_baseline:
            # the reference's op sequence on this GPU with the per-pixel 3x3 / 4x4 products written element-wise (opt-in: its
            # literal form issues ONE batched GEMM of 18.9 M (3x3)(3x1) products, inside which round 1's run aborted with a GPU
            # memory access fault -- DESIGN.md section 6)
            with torch.no_grad(), Of.elementwise_mm():
                fa = (dmaps[0][None, None], dmaps[1:][None, :, None], fcams[0][None], fcams[1:][None], fconf[None], 0.3, 2, 4.0, 1.3)
                tg = timed(lambda: Of.fusion_dynamic_filter(*fa), n=3)
            result["fusion_filter"]["torch_rocm_composite"] = {"ms_per_ref_view": tg, "speedup_of_hip_path": tg / hip_ms,
                                                               "note": "oracle restatement of misc/fusion.py on the GPU, element-wise matrix products"}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            # the reference's op sequence (oracle restatement of misc/fusion.py) on the host cores, quarter of the pixels
            hq, wq = H // 2, W // 2
            dq, cq = synth.synth_depth_maps(hq, wq, Vf + 1, seed=4)
            with torch.no_grad():
                t0 = time.perf_counter()
                Of.fusion_dynamic_filter(dq[0][None, None], dq[1:][None, :, None], cq[0][None], cq[1:][None], torch.rand(1, hq, wq),
                                         0.3, 2, 4.0, 1.3)
                tq = time.perf_counter() - t0
            result["fusion_filter"]["cpu_reference_ops"] = {"ms_per_ref_view": tq * 1e3, "size": f"{wq}x{hq} (a quarter of the pixels)",
                                                            "cores": torch.get_num_threads(), "kind": "port"}
        # DTU branch of the same row (test_dtu_dypcd.py:164-333; parity unpinned -- cv2 is absent): integer pixel grid, double-precision
        # projection chain, cv2.remap's 1/32-pixel bilinear sampling, ten threshold pairs; the reference runs it with numpy on the host
        ddm, dcams = synth.synth_depth_maps(H, W, Vf + 1, seed=4, noise_mm=0.03, pixel_center=0.0)
        ddm, dcams = ddm.to(dev), dcams.to(dev)
        with torch.no_grad():
            dtu_ms = timed(lambda: ops.fusion_dtu_filter(ddm[0].contiguous(), ddm[1:].contiguous(), dcams[0].contiguous(), dcams[1:].contiguous(),
                                                         fconf[:H // 2, :W // 2].contiguous(), 0.5), n=10)
        result["fusion_filter_dtu"] = {"views_per_s": 1e3 / dtu_ms, "ms_per_ref_view": dtu_ms, "src_views": Vf,
                                       "note": f"{W}x{H} depth maps, {Vf} source views, thresholds s=1..10 (dist_base 0.5, diff_base 0.25), one "
                                               "fused kernel per reference view; PARITY UNPINNED (no cv2 here, no reference fixtures)"}
        if not args.no_cpu_baseline:
            from oracle import effi_dtu_filter_oracle as Od
            hq, wq = H // 4, W // 4
            dq, cq = synth.synth_depth_maps(hq, wq, Vf + 1, seed=4, noise_mm=0.03, pixel_center=0.0)
            Kq = [cq[v, 1, :3, :3].numpy() for v in range(Vf + 1)]
            Eq = [cq[v, 0].numpy() for v in range(Vf + 1)]
            t0 = time.perf_counter()
            Od.filter_depth_arrays(dq[0].numpy(), Kq[0], Eq[0], [dq[v].numpy() for v in range(1, Vf + 1)], Kq[1:], Eq[1:],
                                   torch.rand(hq // 2, wq // 2).numpy())
            tq = time.perf_counter() - t0
            result["fusion_filter_dtu"]["cpu_reference_ops"] = {"ms_per_ref_view": tq * 1e3, "size": f"{wq}x{hq} (1/16 of the pixels)", "cores": 1,
                                                                "kind": "port", "note": "numpy restatement of the reference's filter (its remap restated)"}
        del dmaps, ddm

    # ---- secondary (rank 0, N = 1): scope row n4, the producer of the path's inputs -- decoded 8-bit images resident in HBM ->
    # [N,3,H,W] fp32 planes (/255, bilinear resize as scale_mvs_input does it, HWC -> CHW; datasets/general_eval.py:83-117,189)
    if rank == 0 and world == 1 and not args.no_whole_forward:
        src_h = H + 16                      # DTU: 1200 -> 1184 rows (general_eval.py:104-109); same 16-row difference elsewhere
        raws = [torch.randint(0, 256, (src_h, W, 3), dtype=torch.uint8, device=dev) for _ in range(N)]
        planes = torch.empty(N, 3, H, W, device=dev)
        prep_ms = timed(lambda: [ops.image_prepare(raws[v], H, W, out=planes[v]) for v in range(N)], n=20)
        pbytes = N * (3.0 * src_h * W + 12.0 * H * W)
        result["input_prepare"] = {"views_per_s": 1e3 / prep_ms, "ms_per_ref_view": prep_ms, "images": N,
                                   "algorithmic_GBps": pbytes / prep_ms / 1e6,
                                   "note": f"{N} decoded {W}x{src_h} uint8 images in HBM -> [N,3,{H},{W}] fp32, one kernel per image"}
        if not args.no_cpu_baseline:
            from oracle import effi_io_oracle as Oio
            raw_c = raws[0].cpu().numpy()
            t0 = time.perf_counter()
            Oio.resize_linear(raw_c.astype(np.float32) / 255., W, H).transpose(2, 0, 1).copy()
            t1 = time.perf_counter() - t0
            result["input_prepare"]["cpu_reference_ops"] = {"ms_per_ref_view": t1 * N * 1e3, "cores": 1, "kind": "port",
                                                            "sample": "one image timed, scaled to N (numpy restatement of /255 + cv2.resize + transpose)"}
        del raws, planes

    # ---- secondary (rank 0, N = 1): scope row n2, first piece -- forward + backward of the stage-1 warp + correlation
    # (effi_mvs_plus_amd.autograd.warp_correlate) against torch autograd through the reference's grid_sample formulation
    if rank == 0 and world == 1 and not args.no_whole_forward:
        from effi_mvs_plus_amd import autograd as A
        from oracle import effi_oracle as Ob
        f0, c0, p0, d0 = inputs[0]
        st1 = [fv["stage1"][0].contiguous() for fv in f0]
        C1, h1, w1 = st1[0].shape
        D1 = int(nd.split(",")[0])
        pairs1 = p0["stage1"][0].contiguous()
        hyp1, _ = ops.stage1_hypotheses(d0[0].contiguous(), D1)
        gsim = torch.randn(N - 1, D1, h1, w1, device=dev)

        def hip_fb():
            leaves = [x.detach().requires_grad_(True) for x in st1]
            sim = A.warp_correlate(leaves[0], leaves[1:], pairs1, hyp1)
            sim.backward(gsim)
            return leaves[0].grad

        def torch_fb():
            leaves = [x.detach().unsqueeze(0).requires_grad_(True) for x in st1]
            P = [Ob.compose_projection(pairs1[v:v + 1]) for v in range(N)]
            dvals = hyp1.view(1, D1)
            sims = []
            for v in range(1, N):
                wv = Ob.homo_warping_new(leaves[v], P[v], P[0], dvals).view(1, C1, D1, h1, w1)
                sims.append((wv * leaves[0].unsqueeze(2)).mean(1)[0])
            torch.stack(sims).backward(gsim)
            return leaves[0].grad[0]

        try:
            hip_ms = timed(hip_fb, n=10)
            ref_ms = timed(torch_fb, n=3)
            gd = (hip_fb() - torch_fb()).abs().max().item() / max(torch_fb().abs().max().item(), 1e-30)
            with ops.options(warp_lds_kb=-1):                   # the round-1 kernels (direct gather, global atomics), same call
                direct_ms = timed(hip_fb, n=5)
            result["warp_correlate_fwd_bwd"] = {"ms": hip_ms, "ms_torch_rocm_autograd": ref_ms, "speedup": ref_ms / hip_ms,
                                                "ms_direct_kernels_global_atomics": direct_ms,
                                                "grad_ref_max_diff_rel_to_peak": gd,
                                                "note": f"stage-1 shape: C={C1}, {w1}x{h1}, D={D1}, {N - 1} source views; forward + backward "
                                                        "to reference and source features"}
        except Exception as exc:
            result["warp_correlate_fwd_bwd"] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    # ---- baselines (rank 0, N = 1 only): bounded samples of the same workload ------------------------
    if rank == 0 and world == 1:
        from oracle import effi_oracle as O
        f, c, p, d = inputs[0]
        hip_ms = dt / args.steps * 1e3
        if result.get("single_stream") and "ms_per_view" in result["single_stream"]:
            hip_ms = result["single_stream"]["ms_per_view"]       # the baselines run one view at a time: compare with the HIP path doing the same
        if args.torch_baseline_views > 0:
            # reference-style composite path: the oracle's op-for-op torch sequence on this GPU, (a) sync-free (without the
            # reference's NaN probe and torch.unique assert) and (b) literal: with those 34 host synchronisations per view
            # (models/module.py:331-332 x12, models/Effi_MVS_plus.py:109 x22) exactly where the reference has them
            sd_dev = {k: v.to(dev) for k, v in sd.items()}

            def time_composite():
                O.hot_path(sd_dev, f, c, p, d)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.torch_baseline_views):
                    O.hot_path(sd_dev, f, c, p, d)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / args.torch_baseline_views

            with torch.no_grad():
                tb = time_composite()
                with O.literal_syncs():
                    tl = time_composite()
                    n_syncs = O.SYNC_COUNT // (args.torch_baseline_views + 1)
            result["torch_rocm_composite"] = {"value": 1.0 / tb, "unit": "views/s", "ms_per_view": tb * 1e3,
                                              "speedup_of_hip_path": tb * 1e3 / hip_ms,
                                              "sample": f"{args.torch_baseline_views} views, same inputs, stock PyTorch-ROCm ops, sync-free restatement"}
            result["torch_rocm_literal_with_syncs"] = {"value": 1.0 / tl, "unit": "views/s", "ms_per_view": tl * 1e3,
                                                       "host_syncs_per_view": n_syncs, "speedup_of_hip_path": tl * 1e3 / hip_ms,
                                                       "sample": f"{args.torch_baseline_views} views, same inputs, stock PyTorch-ROCm ops with the "
                                                                 "reference's NaN probe and torch.unique assert in place"}
            del sd_dev
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline:
            fc = [{k: v.cpu() for k, v in x.items()} for x in f]
            cc = {k: v.cpu() for k, v in c.items()}
            pc = {k: v.cpu() for k, v in p.items()}
            dc = d.cpu()
            with torch.no_grad():
                t0 = time.perf_counter()
                for _ in range(args.cpu_views):
                    want = O.hot_path(sd, fc, cc, pc, dc)
                tc = (time.perf_counter() - t0) / args.cpu_views
            result["cpu_baseline"] = {"value": 1.0 / tc, "unit": "views/s", "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": f"{args.cpu_views} reference views of the same workload ({args.workload}), hot path only, "
                                                f"oracle/effi_oracle.py (bitwise equal to the reference's CPU PyTorch path), "
                                                f"{tc:.2f} s per view"}
            # parity of the timed workload at its own size: the HIP path's outputs for the same view against the oracle pass that
            # was just timed (normalised by the depth range, the units of SURVEY.md section 8(d); gates: mean <= 1e-3, p99 <= 5e-3)
            with torch.no_grad():
                got = step(0)
                torch.cuda.synchronize()
            rng = synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM

            def dist_(a, b, scale):
                e = ((a.detach().cpu() - b).abs() / scale).flatten()
                return {"mean_norm": float(e.mean()), "p99_norm": float(e.kthvalue(max(1, int(0.99 * e.numel()))).values),
                        "max_norm": float(e.max())}

            per = [dist_(a, b, rng) for a, b in zip(got["depth"], want["depth"])]
            cf = dist_(got["photometric_confidence"], want["photometric_confidence"], 1.0)
            result["parity_vs_oracle"] = {
                "depth": per, "photometric_confidence": cf,
                "worst_depth_mean_norm": max(x["mean_norm"] for x in per), "worst_depth_p99_norm": max(x["p99_norm"] for x in per),
                "gate": {"mean_norm": 1e-3, "p99_norm": 5e-3, "confidence_mean_abs": 1e-3},
                "pass": bool(max(x["mean_norm"] for x in per) <= 1e-3 and max(x["p99_norm"] for x in per) <= 5e-3 and cf["mean_norm"] <= 1e-3),
                "note": f"{len(per)} depth maps + confidence of the timed workload ({args.workload}, view 0, precision {precision}) vs "
                        "oracle/effi_oracle.py on the host; |d - d_ref| / (depth_max - depth_min)"}
            if bf16_outputs is not None:        # parity of the bf16-operand variant timed above (own tolerance: final depth <= 1e-2)
                perb = [dist_(a_, b_, rng) for a_, b_ in zip(bf16_outputs, want["depth"])]
                result["bf16_operands"]["parity_vs_oracle"] = {
                    "final_depth": perb[-1], "worst_depth_mean_norm": max(x["mean_norm"] for x in perb),
                    "worst_depth_p99_norm": max(x["p99_norm"] for x in perb), "gate": {"final_depth_mean_norm": 1e-2},
                    "pass": bool(perb[-1]["mean_norm"] <= 1e-2)}
    n_differing = 0
    if rank == 0:
        print(json.dumps(result))
        ss = result.get("single_stream") or {}
        n_differing = int(ss.get("timed_in_flight_views_differing_from_single_stream") or 0)
    if distributed:
        dist.destroy_process_group()
    if n_differing:
        # a view produced while other views were in flight differs from the same view computed alone: the number above is void
        print(f"[bench] {n_differing} of the timed in-flight views differ bitwise from the single-stream result", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
