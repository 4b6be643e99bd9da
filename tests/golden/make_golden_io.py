#!/usr/bin/env python3
"""Golden vectors for scope row n4 (PFM files), produced by RUNNING the reference's own datasets/data_io.py (save_pfm,
read_pfm) in the build container.  data_io.py imports torchvision (line 4, for get_transform) and cv2 (line 128, for the
training-time crop classes); neither is installed, so empty stand-in modules are registered for the import only -- none of
the functions used here touch them.  The reference's dataset classes (general_eval.py, tank.py) import cv2, which is absent: they cannot be run,
and no vectors exist for them (oracle/effi_io_oracle.py says so).  Run from the repo root: python tests/golden/make_golden_io.py
"""
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
tv = types.ModuleType("torchvision")
tv.transforms = types.ModuleType("torchvision.transforms")
sys.modules.setdefault("torchvision", tv)
sys.modules.setdefault("torchvision.transforms", tv.transforms)
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, "/root/reference")
import datasets.data_io as ref_io  # noqa: E402  (the reference)


def main():
    rng = np.random.default_rng(13)
    cases = {"gray": rng.standard_normal((5, 7)).astype(np.float32) * 300,
             "gray1": rng.random((4, 6, 1)).astype(np.float32),
             "color": rng.random((3, 5, 3)).astype(np.float32),
             "depth": (425 + 510 * rng.random((37, 50))).astype(np.float32)}
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for name, img in cases.items():
            for scale in (1, 2.5):
                p = os.path.join(d, f"{name}.pfm")
                ref_io.save_pfm(p, img, scale)
                raw = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
                back, sc = ref_io.read_pfm(p)
                tag = f"{name}_s{str(scale).replace('.', 'p')}"
                out[f"{tag}_image"] = img
                out[f"{tag}_file"] = raw
                out[f"{tag}_read"] = np.ascontiguousarray(back)
                out[f"{tag}_scale"] = np.float64(sc)
        # a big-endian file (positive scale), as other tools write them
        be = rng.random((4, 3)).astype(np.float32)
        p = os.path.join(d, "be.pfm")
        with open(p, "wb") as f:
            f.write(b"Pf\n3 4\n1.000000\n")
            f.write(np.flipud(be).astype(">f4").tobytes())
        back, sc = ref_io.read_pfm(p)
        out["be_file"] = np.frombuffer(open(p, "rb").read(), dtype=np.uint8)
        out["be_read"] = np.ascontiguousarray(back).astype(np.float32)
        out["be_scale"] = np.float64(sc)
    np.savez_compressed(os.path.join(HERE, "g13_io.npz"), **out)
    print("wrote g13_io.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
