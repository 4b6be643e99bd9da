#!/usr/bin/env python3
"""Host-side profile (cProfile) of the eager training step: where the ~12 us per launch go.  usage: prof_train_host.py [n_steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from common import build_model  # noqa: E402
from effi_mvs_plus_amd import synth  # noqa: E402
from effi_mvs_plus_amd.models import mvs_loss  # noqa: E402
import bench_train as BT  # noqa: E402

DEV = "cuda:0"
H, W, N = 512, 640, 5
net, _ = build_model("48,8,8", seed=2, device=DEV)
imgs, pm, dv = synth.synth_sample(H, W, N, seed=10)
imgs, pm, dv = imgs.to(DEV), {k: v.to(DEV) for k, v in pm.items()}, dv.to(DEV)
gt, mask = BT.loss_inputs(H, W, 1, 1)
net.train()
opt = torch.optim.AdamW(net.parameters(), lr=1e-5)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = mvs_loss(net(imgs, pm, dv)["depth"], gt, mask, BT.DLOSS)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
