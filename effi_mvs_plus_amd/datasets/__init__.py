"""Scope row n4: the input pipeline and on-disk formats on either side of the hot path (reference: datasets/).

``data_io``      PFM reader / writer (datasets/data_io.py:61-126)
``general_eval`` DTU-style evaluation dataset (datasets/general_eval.py:8-228)
``tank``         Tanks-and-Temples evaluation dataset (datasets/tank.py:13-183)

Parsing and matrix bookkeeping are host code like the reference's; the per-view image arithmetic (/255, bilinear resize,
HWC -> CHW) is one HIP kernel (``ops.image_prepare``), so ``__getitem__`` needs the GPU and fails loudly without it.
"""
