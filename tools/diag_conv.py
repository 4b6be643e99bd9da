import sys, os, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.nn.functional as F
from effi_mvs_plus_amd import autograd as A, ops
from test_gpu_train import rel, leaf
DEV = "cuda:0"
for ks, cin, cout, B, h, w, act in ((1, 96, 36, 2, 16, 20, "none"), (3, 48, 96, 2, 16, 20, "relu"), (3, 48, 96, 1, 16, 20, "relu"), (1, 96, 36, 1, 16, 20, "none"),
                                    (3, 48, 48, 2, 16, 20, "relu"), (3, 96, 96, 2, 16, 20, "sigmoid")):
    g = torch.Generator().manual_seed(1)
    W = torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks)
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, h, w, generator=g)
    gy = torch.randn(B, cout, h, w, generator=g)
    f = {"none": lambda v: v, "relu": F.relu, "sigmoid": torch.sigmoid}[act]
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "sigmoid": ops.ACT_SIGMOID}[act]
    Wc, bc, xc = leaf(W), leaf(b), leaf(x)
    want = f(F.conv2d(xc, Wc, bc, padding=ks // 2)); want.backward(gy)
    Wd, bd, xd = leaf(W, DEV), leaf(b, DEV), leaf(x, DEV)
    got = A.conv2d([xd], Wd, bd, code); got.backward(gy.to(DEV))
    print(ks, cin, cout, "B", B, "fwd", f"{rel(got, want):.1e}", "dW", f"{rel(Wd.grad, Wc.grad):.1e}", "db", f"{rel(bd.grad, bc.grad):.1e}", "dx", f"{rel(xd.grad, xc.grad):.1e}")
