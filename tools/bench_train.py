#!/usr/bin/env python3
"""One training step (forward in train mode -> mvs_loss -> backward -> AdamW step, train.py:229-263) at a DTU training shape:
the HIP training path of this package against torch autograd through the oracle's op sequence with stock PyTorch-ROCm operators
(MIOpen convolutions, grid_sample) on the same GPU, same weights, same sample.   usage: bench_train.py [H W N B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import build_model  # noqa: E402
from effi_mvs_plus_amd import synth  # noqa: E402
from effi_mvs_plus_amd.models import mvs_loss  # noqa: E402

DEV = "cuda:0"
DLOSS = [1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4]           # train.py:246: stage of each of the 13 depth maps


def loss_inputs(H, W, B, seed):
    g = torch.Generator().manual_seed(seed)
    gt, mask = {}, {}
    for k, f in (("stage1", 8), ("stage2", 4), ("stage3", 2), ("stage4", 1)):
        gt[k] = (synth.DEPTH_MIN_MM + (synth.DEPTH_MAX_MM - synth.DEPTH_MIN_MM) * torch.rand(B, H // f, W // f, generator=g)).to(DEV)
        mask[k] = (torch.rand(B, H // f, W // f, generator=g) > 0.3).float().to(DEV)
    return gt, mask


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    H, W, N, B = (int(v) for v in (sys.argv[1:5] + ["512", "640", "5", "1"][len(sys.argv) - 1:]))
    nd = "48,8,8"
    net, sd = build_model(nd, seed=2, device=DEV)
    samples = [synth.synth_sample(H, W, N, seed=10 + b) for b in range(B)]
    imgs = torch.cat([s_[0] for s_ in samples]).to(DEV)
    pm = {k: torch.cat([s_[1][k] for s_ in samples]).to(DEV) for k in samples[0][1]}
    dv = torch.cat([s_[2] for s_ in samples]).to(DEV)
    gt, mask = loss_inputs(H, W, B, 1)

    net.train()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-5)

    def hip_step():
        opt.zero_grad(set_to_none=True)
        loss, _ = mvs_loss(net(imgs, pm, dv)["depth"], gt, mask, DLOSS)
        loss.backward()
        opt.step()
        return loss

    hip_ms = timed(hip_step, 5)
    print(f"HIP training step {W}x{H} N={N} B={B}: {hip_ms:.1f} ms  ({B * 1e3 / hip_ms:.1f} samples/s)")

    if not os.environ.get("BENCH_TRAIN_NO_GRAPH"):
        # the same step (static loss, capturable AdamW) captured into one HIP graph and replayed: no per-launch host work
        from effi_mvs_plus_amd import train_graph
        opt_g = torch.optim.AdamW(net.parameters(), lr=1e-5, capturable=True)
        step = train_graph.GraphedTrainStep(net, opt_g, imgs, pm, dv, gt, mask)
        graph_ms = timed(lambda: step(), 10)
        print(f"HIP training step, graph replay (train_graph.GraphedTrainStep): {graph_ms:.1f} ms  ({B * 1e3 / graph_ms:.1f} samples/s)")
        del step

    if os.environ.get("BENCH_TRAIN_HIP_ONLY"):
        return
    # stock PyTorch-ROCm: autograd through the oracle's training-mode op sequence, leaves = one tensor per parameter
    from oracle import effi_oracle as O
    sd2 = {k: v.detach().clone().to(DEV) for k, v in net.state_dict().items()}
    groups = {}
    for k, p_ in net.named_parameters(remove_duplicate=False):
        groups.setdefault(id(p_), []).append(k)
    leaves = []
    for ks in groups.values():
        lf = sd2[ks[0]].requires_grad_(True)
        leaves.append(lf)
        for k in ks:
            sd2[k] = lf
    opt2 = torch.optim.AdamW(leaves, lr=1e-5)

    def torch_step():
        opt2.zero_grad(set_to_none=True)
        with O.training(0.0):
            out = O.full_forward(sd2, imgs, pm, dv, ndepths=tuple(int(v) for v in nd.split(",")))
            loss, _ = O.mvs_loss(out["depth"], gt, mask, DLOSS)
        loss.backward()
        opt2.step()
        return loss

    try:
        ref_ms = timed(torch_step, 3)
        print(f"PyTorch-ROCm autograd through the oracle: {ref_ms:.1f} ms  -> HIP path {ref_ms / hip_ms:.2f}x")
    except Exception as exc:      # noqa: BLE001
        print(f"PyTorch-ROCm reference step failed: {type(exc).__name__}: {exc}")


if __name__ == "__main__":
    main()
