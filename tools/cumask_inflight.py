#!/usr/bin/env python3
"""Views in flight on CU-PARTITIONED streams (experiment).

The headline figure keeps 3 views in flight on 3 ordinary streams: a kernel that fills every CU's workgroup slots leaves the other
views' kernels only its tail, so the gain over one view after the other is ~10 %.  Here each in-flight view gets its own stream
created with hipExtStreamCreateWithCUMask -- a disjoint share of the 256 CUs -- so that the latency-bound low-resolution kernels of
one view (which cannot fill 256 CUs anyway) run NEXT TO the other views' kernels instead of in front of them.

usage: tools/cumask_inflight.py [workload]     env: PARTS="1,2,3,4,8"  PATTERNS="none,block,stride"  STEPS=60
  none   = ordinary streams (the bench's form);  block = stream k owns mask bits [k n/K, (k+1) n/K);  stride = bits k, k+K, ...
Every replay's final depth map is compared bitwise with the single-stream result of the same slot."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_model  # noqa: E402
from effi_mvs_plus_amd import ops, synth  # noqa: E402
from effi_mvs_plus_amd.graph import HotPathGraph  # noqa: E402

WL = {"cfg3": (1184, 1600, 5, "48,8,8"), "cfg2": (576, 800, 5, "48,8,8"), "cfg4": (1056, 1920, 7, "48,8,8")}
N_CU = 256


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError("libamdhip64 is not loaded")


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * (N_CU // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), N_CU // 32, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    H, W, N, nd = WL[wl]
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net, _ = build_model(nd, seed=1, device=dev)
    inputs = []
    with torch.no_grad():
        for i in range(2):
            imgs, pm, dv = synth.synth_sample(H, W, N, seed=i)
            imgs = imgs.to(dev)
            feats = [net.feature(imgs[:, v]) for v in range(N)]
            ctx = net.cnet_depth(imgs[:, 0])
            inputs.append((feats, ctx, {k: v.to(dev) for k, v in pm.items()}, dv.to(dev)))
    torch.cuda.synchronize()
    hip = hip_runtime()
    steps = int(os.environ.get("STEPS", "60"))
    max_k = max(int(k) for k in os.environ.get("PARTS", "1,2,3,4,8").split(","))
    ops.set_branches(os.environ.get("BRANCHES", "0") == "1")   # 0: linear graphs -- every node of a replay runs on the stream it is launched into
    graphed = HotPathGraph(net, *inputs[0], slots=max_k)
    for i in range(max_k):
        graphed.load(i, *inputs[i % 2])
    torch.cuda.synchronize()
    ref = []
    for i in range(max_k):
        ref.append(graphed.replay(i)["depth"][-1].clone())
    torch.cuda.synchronize()

    def run(streams):
        K = len(streams)
        outs, confs = [], []
        cur = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(cur)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stagger = float(os.environ.get("STAGGER_MS", "0"))       # lane k starts k * STAGGER_MS late (inside the timed region)
        if stagger > 0:
            for k in range(1, K):
                with torch.cuda.stream(streams[k]):
                    torch.cuda._sleep(int(k * stagger * 1e-3 * 100e6))     # device-side spin, ~100 MHz counter
        for i in range(steps):
            with torch.cuda.stream(streams[i % K]):
                out = graphed.replay(i % K)
                outs.append(out["depth"][-1].clone())
                if os.environ.get("CLONE_CONF", "0") == "1":
                    confs.append(out["photometric_confidence"].clone())
        for st in streams:
            cur.wait_stream(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bad = sum(int(not torch.equal(o, ref[i % K])) for i, o in enumerate(outs))
        return steps / dt, bad

    for K in [int(k) for k in os.environ.get("PARTS", "1,2,3,4,8").split(",")]:
        for pat in os.environ.get("PATTERNS", "none,block,stride").split(","):
            if pat == "none":
                streams = [torch.cuda.Stream() for _ in range(K)]
            elif pat == "block":
                streams = [masked_stream(hip, range(k * N_CU // K, (k + 1) * N_CU // K)) for k in range(K)]
            else:
                streams = [masked_stream(hip, range(k, N_CU, K)) for k in range(K)]
            run(streams)
            res = [run(streams) for _ in range(2)]
            print(f"{wl} K={K} {pat:6s}: " + "  ".join(f"{v:7.1f} views/s ({1e3 / v:.3f} ms/view, {b} differing)" for v, b in res), flush=True)


if __name__ == "__main__":
    main()
