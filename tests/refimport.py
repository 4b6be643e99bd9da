"""Import the reference's ``models`` package on CPU (build container only).

Recipe from SURVEY.md section 8(c): ``models/module.py:7`` imports ``utils`` which needs torchvision (absent),
so a stub ``utils`` module is registered first; bytecode writing is disabled because the reference
tree is read-only.  Returns None when /root/reference does not exist (e.g. on the GPU box).
"""
import argparse
import os
import sys
import types

REF_ROOT = "/root/reference"


def load_reference():
    if not os.path.isdir(os.path.join(REF_ROOT, "models")):
        return None
    sys.dont_write_bytecode = True
    if "utils" not in sys.modules or not hasattr(sys.modules["utils"], "local_pcd"):
        stub = types.ModuleType("utils")
        stub.local_pcd = lambda *a, **k: None
        sys.modules["utils"] = stub
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import models  # noqa: F401  (the reference's package)
    return types.SimpleNamespace(
        main=sys.modules["models.Effi_MVS_plus"],
        module=sys.modules["models.module"],
        update=sys.modules["models.update"],
    )


def reference_model(ndepths="48,8,8", gru_iters="3,3,3", cost_num=3):
    ref = load_reference()
    if ref is None:
        return None, None
    args = argparse.Namespace(ndepths=ndepths, GRUiters=gru_iters, CostNum=cost_num)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = ref.main.Effi_MVS_plus(args)
    net.eval()
    return ref, net
