import torch, sys
sys.path.insert(0, "/root/repo")
from effi_mvs_plus_amd import ops, packing
dev = "cuda:0"
for (h, w, co) in [(592, 800, 16), (296, 400, 32), (148, 200, 48)]:
    x = torch.rand(1, h, w, device=dev)
    wt = torch.randn(co, 1, 7, 7, device=dev) * 0.1
    b = torch.randn(co, device=dev) * 0.1
    w7, b7 = packing.pack_conv2d_c1k7(wt, b)
    out = torch.empty(co, h, w, device=dev)
    for _ in range(3):
        ops.conv2d_c1k7_relu(x, w7, b7, co, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv2d_c1k7_relu(x, w7, b7, co, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"c1k7 {h}x{w} cout {co}: {us:.1f} us  ({4.0 * h * w * (co + 1) / us / 1e3:.0f} GB/s)")
